// Header-only C++ adapter with the reference's method names: a ROS build elsewhere can give its
// `class Localization` (reference src/localization/localization.h:99-128) this member instead of the g2o optimizer and
// forward each callback's numeric fields.  No ROS types here, so it compiles standalone; the ROS-typed overloads are
// one-liners on the caller's side.
#pragma once
#include <array>
#include <cstdio>
#include <ctime>
#include <stdexcept>
#include <string>
#include <vector>

#include "localization_amd.h"

namespace localization_amd {

struct PoseStampedLike {            // geometry_msgs::PoseStamped without ROS
    double stamp = 0;
    std::array<double, 3> position{{0, 0, 0}};
    std::array<double, 4> orientation_xyzw{{0, 0, 0, 1}};
};

class Localization {
public:
    // mirrors the parameters the reference constructor reads (localization.cpp:58-159)
    struct Params {
        int trajectory_length = 0;
        double maximum_velocity = 1.0, distance_outlier = 1.0, minimum_optimize_error = 1000.0;
        int maximum_iteration = 20;
        bool publish_range = false, publish_pose = false, publish_twist = false, publish_lidar = false, publish_imu = false;
        bool relative_range_topic = false;
        bool numeric_jacobian = true;         // g2o's central-difference range Jacobians = the reference's exact configuration (false: analytic, the opt-in fast mode)
        std::vector<int> nodesId;            // /uwb/nodesId, last = the moving tag
        std::vector<double> nodesPos;        // /uwb/nodesPos
        std::vector<double> antennaOffset;   // /uwb/antennaOffset (may be empty)
        int device = 0;
    };

    explicit Localization(const Params& p) : trajectory_length_(p.trajectory_length) {
        loc_node_config c;
        loc_node_default_config(&c);
        c.trajectory_length = p.trajectory_length; c.maximum_velocity = p.maximum_velocity;
        c.distance_outlier = p.distance_outlier; c.maximum_iteration = p.maximum_iteration;
        c.minimum_optimize_error = p.minimum_optimize_error;
        c.publish_range = p.publish_range; c.publish_pose = p.publish_pose; c.publish_twist = p.publish_twist;
        c.publish_lidar = p.publish_lidar; c.publish_imu = p.publish_imu; c.has_relative_range = p.relative_range_topic;
        c.jacobian = p.numeric_jacobian ? LOC_JAC_NUMERIC_G2O : LOC_JAC_ANALYTIC;
        std::vector<int32_t> ids(p.nodesId.begin(), p.nodesId.end());
        self_id_ = ids.empty() ? 0 : ids.back();
        const int rc = loc_node_create(&node_, p.device, &c, (int32_t)ids.size(), ids.data(), p.nodesPos.data(),
                                       (int32_t)(p.antennaOffset.size() / 3), p.antennaOffset.empty() ? nullptr : p.antennaOffset.data());
        if (rc != LOC_OK) throw std::runtime_error(std::string("localization_amd: ") + loc_last_error());
    }
    // Localization::~Localization (localization.cpp:708-717): with logging on, path->poses[T/2 .. T-1] goes to the optimized log
    ~Localization() {
        if (flag_save_file_ && node_) {
            for (const PoseStampedLike& p : flushTail()) save_file(p, optimized_filename_);
            std::printf("Results Loged to file: %s\n", optimized_filename_.c_str());
        }
        loc_node_destroy(node_);
    }
    Localization(const Localization&) = delete;
    Localization& operator=(const Localization&) = delete;

    // void addRangeEdge(const uwb_driver::UwbRange::ConstPtr&)            localization.cpp:297
    bool addRangeEdge(int requester_id, int responder_id, double stamp, float distance, float distance_err, int antenna,
                      const std::string& frame_id) {
        return handle(loc_node_add_range(node_, requester_id, responder_id, stamp, distance, distance_err, antenna, frame_id.c_str(), &last_));
    }
    // void addImuEdge(const sensor_msgs::Imu::ConstPtr&)                   localization.cpp:499
    bool addImuEdge(double stamp, const std::array<double, 4>& orientation_xyzw, const std::array<double, 9>& orientation_covariance,
                    const std::string& frame_id) {
        return handle(loc_node_add_imu(node_, stamp, orientation_xyzw.data(), orientation_covariance.data(), frame_id.c_str(), &last_));
    }
    // void addPoseEdge(const geometry_msgs::PoseWithCovarianceStamped::ConstPtr&)    localization.cpp:254
    bool addPoseEdge(double stamp, const std::array<double, 7>& pose_xyz_qxyzw, const std::array<double, 36>& covariance,
                     const std::string& frame_id) {
        return handle(loc_node_add_pose(node_, stamp, pose_xyz_qxyzw.data(), covariance.data(), frame_id.c_str(), &last_));
    }
    // void addTwistEdge(const geometry_msgs::TwistWithCovarianceStamped::ConstPtr&)  localization.cpp:438
    bool addTwistEdge(double stamp, const std::array<double, 6>& twist_lin_ang, const std::array<double, 36>& covariance,
                      const std::string& frame_id = "") {
        return handle(loc_node_add_twist(node_, stamp, twist_lin_ang.data(), covariance.data(), frame_id.c_str(), &last_));
    }
    // void addLidarEdge(const geometry_msgs::PoseWithCovarianceStamped::ConstPtr&)   localization.cpp:462
    bool addLidarEdge(double stamp, double z, const std::string& frame_id) {
        return handle(loc_node_add_lidar(node_, stamp, z, frame_id.c_str(), &last_));
    }
    // void addRLRangeEdge(const uwb_reloc::uwbTalkData::ConstPtr&)        localization.cpp:378 (built with -DRELATIVE_LOCALIZATION, CMakeLists.txt:137)
    // uwbTalkData: time_stamp, rqstrId, rspdrId, d, rqstr_vx / vy / vz
    bool addRLRangeEdge(double time_stamp, int rqstrId, int rspdrId, double d, const std::array<double, 3>& rqstr_velocity) {
        return handle(loc_node_add_rl_range(node_, rqstrId, rspdrId, time_stamp, d, rqstr_velocity.data(), &last_));
    }
    // void solve(); void publish();                                        localization.cpp:164, :195
    bool solve() { return handle(loc_node_solve(node_, &last_)); }

    // void set_file()                                                      localization.cpp:645-706: "<prefix>_realtime<stamp>.txt" and
    // "<prefix>_optimized<stamp>.txt" with the reference's header lines; from then on every solve that publishes appends the realtime pose
    // and path[T/2] (publish(), :221-225), and the destructor appends path[T/2 .. T-1].  stamp_suffix empty: "_%Y_%b_%d_%H_%M_%S.txt" of now.
    void set_file(const std::string& name_prefix, const Params& p, const std::string& stamp_suffix = "") {
        std::string s = stamp_suffix;
        if (s.empty()) {
            char buf[40];
            const std::time_t now = std::time(nullptr);
            std::strftime(buf, sizeof buf, "_%Y_%b_%d_%H_%M_%S.txt", std::localtime(&now));
            s = buf;
        }
        realtime_filename_ = name_prefix + "_realtime" + s;
        optimized_filename_ = name_prefix + "_optimized" + s;
        for (const std::string& fn : {realtime_filename_, optimized_filename_}) {
            std::FILE* f = std::fopen(fn.c_str(), "w");
            if (!f) throw std::runtime_error("localization_amd: cannot open " + fn);
            if (p.antennaOffset.empty()) {   // set_file(), :645-671
                std::fprintf(f, "# iteration_max:%d\n# trajectory_length:%d\n# maximum_velocity:%g\n", p.maximum_iteration, p.trajectory_length, p.maximum_velocity);
            } else {                          // set_file(antennaOffset), :673-706: the files are re-opened with ios::trunc, so this line is all that stays
                std::fprintf(f, "# antenna offsets: ");
                for (size_t i = 0; i < p.antennaOffset.size(); ++i) std::fprintf(f, "%g%s", p.antennaOffset[i], i + 1 < p.antennaOffset.size() ? "," : "\n");
            }
            std::fclose(f);
        }
        flag_save_file_ = true;
    }
    const std::string& realtimeFilename() const { return realtime_filename_; }
    const std::string& optimizedFilename() const { return optimized_filename_; }
    // path->poses[T/2 .. T-1]: what the destructor logs (loc_node_flush_tail)
    std::vector<PoseStampedLike> flushTail() const {
        std::vector<double> buf((size_t)trajectory_length_ * 8 + 8);
        const int n = loc_node_flush_tail(node_, buf.data(), trajectory_length_);
        std::vector<PoseStampedLike> out;
        for (int i = 0; i < n; ++i) out.push_back(to_pose(&buf[(size_t)i * 8]));
        return out;
    }

    // what publish() would put on the wire after the last solve
    bool published() const { return last_.published != 0; }
    double chi2() const { return last_.chi2; }
    PoseStampedLike realtimePose() const { return to_pose(last_.realtime); }      // realtime/pose
    PoseStampedLike optimizedPose() const { return to_pose(last_.optimized); }    // optimized/pose = path[T/2]
    std::vector<PoseStampedLike> optimizedPath() const {                          // optimized/path (Robot::vertices2path)
        std::vector<double> buf((size_t)trajectory_length_ * 8);
        const int n = loc_node_get_path(node_, self_id_, buf.data(), trajectory_length_);
        std::vector<PoseStampedLike> out;
        for (int i = 0; i < n; ++i) out.push_back(to_pose(&buf[(size_t)i * 8]));
        return out;
    }

private:
    static PoseStampedLike to_pose(const double* p) {
        PoseStampedLike o;
        o.stamp = p[0]; o.position = {{p[1], p[2], p[3]}}; o.orientation_xyzw = {{p[4], p[5], p[6], p[7]}};
        return o;
    }
    bool handle(int rc) {
        if (rc < 0) throw std::runtime_error(std::string("localization_amd: ") + loc_last_error());  // reference: map::at throws
        if (rc == 1 && flag_save_file_ && last_.published) {   // publish(), localization.cpp:221-225
            save_file(to_pose(last_.realtime), realtime_filename_);
            save_file(to_pose(last_.optimized), optimized_filename_);
        }
        return rc == 1;
    }
    // Localization::save_file, localization.cpp:630-642: "%.9f" stamp, then x y z qx qy qz qw at the stream's default precision (6 digits)
    static void save_file(const PoseStampedLike& p, const std::string& filename) {
        std::FILE* f = std::fopen(filename.c_str(), "a");
        if (!f) return;
        std::fprintf(f, "%.9f %g %g %g %g %g %g %g\n", p.stamp, p.position[0], p.position[1], p.position[2], p.orientation_xyzw[0],
                     p.orientation_xyzw[1], p.orientation_xyzw[2], p.orientation_xyzw[3]);
        std::fclose(f);
    }
    bool flag_save_file_ = false;
    std::string realtime_filename_, optimized_filename_;
    loc_node* node_ = nullptr;
    loc_node_output last_{};
    int trajectory_length_ = 0;
    int self_id_ = 0;
};

}  // namespace localization_amd
