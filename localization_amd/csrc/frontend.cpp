// Host front-end of the window path: the reference's `class Localization` + `class Robot` semantics
// (/root/reference/src/localization/localization.{h,cpp}, robot.{h,cpp}) behind the C ABI `loc_node_*`.
// It keeps the graph the reference would hand to g2o in flat host arrays and, whenever the reference would call
// solve() (localization.cpp:164-170), packs the active part of the graph into one window instance and runs the
// gfx950 window kernel (window_kernel.hip) through loc_window_*.  ROS messages are replaced by their numeric fields.
// There is no CPU solve path: without a HIP device loc_node_create fails.
#include "../../include/localization_amd.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

extern int locamd_fail(int code, const char* what);

namespace {

struct Iso { double R[9]; double t[3]; };

inline Iso iso_identity() { Iso x{}; x.R[0] = x.R[4] = x.R[8] = 1.0; return x; }
inline void mul3(const double* A, const double* B, double* C) {
    double o[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    std::memcpy(C, o, sizeof(o));
}
inline Iso iso_inverse(const Iso& a) {
    Iso o;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = a.R[j * 3 + i];
    for (int i = 0; i < 3; ++i) o.t[i] = -(o.R[i * 3] * a.t[0] + o.R[i * 3 + 1] * a.t[1] + o.R[i * 3 + 2] * a.t[2]);
    return o;
}
// Eigen::Quaterniond(w,x,y,z).toRotationMatrix(), no normalisation (localization.cpp:509 relies on it)
inline void quat_to_R(double w, double x, double y, double z, double* R) {
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Eigen::Quaterniond(R) then tf::poseEigenToMsg's w >= 0 flip; out = (x, y, z, w)
inline void R_to_quat_xyzw(const double* R, double* q) {
    double w, x, y, z, t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t;
        x = (R[7] - R[5]) * t; y = (R[2] - R[6]) * t; z = (R[3] - R[1]) * t;
    } else if (R[0] >= R[4] && R[0] >= R[8]) {
        t = std::sqrt(R[0] - R[4] - R[8] + 1.0); x = 0.5 * t; t = 0.5 / t;
        w = (R[7] - R[5]) * t; y = (R[3] + R[1]) * t; z = (R[6] + R[2]) * t;
    } else if (R[4] > R[0] && R[4] >= R[8]) {
        t = std::sqrt(R[4] - R[8] - R[0] + 1.0); y = 0.5 * t; t = 0.5 / t;
        w = (R[2] - R[6]) * t; z = (R[7] + R[5]) * t; x = (R[1] + R[3]) * t;
    } else {
        t = std::sqrt(R[8] - R[0] - R[4] + 1.0); z = 0.5 * t; t = 0.5 / t;
        w = (R[3] - R[1]) * t; x = (R[2] + R[6]) * t; y = (R[5] + R[7]) * t;
    }
    if (w < 0) { w = -w; x = -x; y = -y; z = -z; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}
// MatrixXd::inverse() of a 6x6 (partial pivoting); false if singular
bool invert6(const double* A, double* out) {
    double M[6][12];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 6; ++c) {
        int p = c;
        for (int r = c + 1; r < 6; ++r) if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (M[p][c] == 0.0 || !std::isfinite(M[p][c])) return false;
        if (p != c) for (int j = 0; j < 12; ++j) std::swap(M[c][j], M[p][j]);
        const double d = M[c][c];
        for (int j = 0; j < 12; ++j) M[c][j] /= d;
        for (int r = 0; r < 6; ++r) {
            if (r == c || M[r][c] == 0.0) continue;
            const double f = M[r][c];
            for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) out[i * 6 + j] = M[i][6 + j];
    return true;
}

struct Header { double stamp = 0.0; std::string frame_id; };

// sensor_type, localization.h:90-97
enum { ST_POSE = 1, ST_RANGE = 2, ST_TWIST = 3, ST_COUNT = 5 };

struct Vertex { Iso est; bool fixed; };
struct RangeEdge { int v0, v1; double meas, info; double off[3]; };
struct PriorEdge { int v; Iso zinv; double info[6]; };
struct Se3Edge { int vi, vj; Iso zinv; double info[36]; bool robust; };

// class Robot (robot.h:60-120): a ring of T vertex slots, ids ID + slot * 300
struct RobotRing {
    int id = 0, T = 1, index = 0;
    bool is_static = true;
    std::vector<Header> header;
    bool type_has[ST_COUNT] = {false, false, false, false, false};
    int type_index[ST_COUNT] = {0, 0, 0, 0, 0};
    bool hdr_has[ST_COUNT] = {false, false, false, false, false};
    Header headers[ST_COUNT];
    int slot_vertex(int slot) const { return id + slot * 300; }  // robot.cpp:43,94
    int last_vertex() const { return slot_vertex(index); }
};

}  // namespace

struct loc_node {
    loc_node_config cfg{};
    int device = 0;
    std::map<int, Vertex> vertices;  // g2o vertex id -> vertex (std::map: ascending id = g2o's Hessian order)
    std::vector<RangeEdge> ranges;
    std::vector<PriorEdge> priors;
    std::vector<Se3Edge> se3s;
    std::vector<RobotRing> robots;
    int self_id = 0;
    std::vector<double> antenna;  // offsets[] translations (xyz each)
    int number_measurements = 0;
    int key_vertex = -1;
    bool deferred = false, pending = false;
    loc_window* win = nullptr;            // cached single-instance solver (anchors are part of its device state)
    std::vector<double> win_anchors;
    loc_window_caps caps{16, 0, 0, 0, -1};     // LIMIT of what pack() accepts: nv_max = the most active poses (set in loc_node_create);
                                               // the edge counts have no limit (the reference has none): the handle grows on demand
    loc_window_caps win_caps{0, 0, 0, 0, 0};   // capacities of the cached handle: what the packed graphs needed so far
    int last_kind = LOC_WINDOW_KERNEL_NONE;                    // LOC_WINDOW_KERNEL_* of the last solve
    double t_pack_ms = 0, t_solve_ms = 0, t_kernel_ms = 0;   // the last solve: host packing, loc_window_solve_host (copies + launch + sync), its kernel

    RobotRing* robot(int id) { for (auto& r : robots) if (r.id == id) return &r; return nullptr; }
    void remove_vertex(int vid) {  // optimizer.removeVertex(v, false): the vertex and every edge touching it
        vertices.erase(vid);
        ranges.erase(std::remove_if(ranges.begin(), ranges.end(), [&](const RangeEdge& e) { return e.v0 == vid || e.v1 == vid; }), ranges.end());
        priors.erase(std::remove_if(priors.begin(), priors.end(), [&](const PriorEdge& e) { return e.v == vid; }), priors.end());
        se3s.erase(std::remove_if(se3s.begin(), se3s.end(), [&](const Se3Edge& e) { return e.vi == vid || e.vj == vid; }), se3s.end());
    }
    // Robot::last_vertex(type), robot.cpp:113-118
    int last_vertex_type(RobotRing& r, int type) {
        if (!r.type_has[type]) { r.type_has[type] = true; r.type_index[type] = r.index; }
        if (!r.hdr_has[type]) { r.hdr_has[type] = true; r.headers[type] = r.header[r.index]; }
        return r.slot_vertex(r.type_index[type]);
    }
    // Robot::new_vertex, robot.cpp:75-110
    int new_vertex(RobotRing& r, int type, const Header& h) {
        if (!r.type_has[type]) { r.type_has[type] = true; r.type_index[type] = r.index; }
        if (!r.hdr_has[type]) { r.hdr_has[type] = true; r.headers[type] = h; }
        if (r.is_static) { r.header[r.index] = h; return last_vertex_type(r, type); }
        const Iso prev = vertices.at(r.last_vertex()).est;  // new pose starts at the previous estimate, :90
        r.index = (r.index + 1) % r.T;
        const int vid = r.slot_vertex(r.index);
        remove_vertex(vid);  // oldest pose leaves with its edges, :96
        vertices[vid] = Vertex{prev, false};
        r.header[r.index] = h;
        r.type_index[type] = r.index;
        r.headers[type] = h;
        return vid;
    }
};

namespace {

void pose_out(const loc_node* n, int vid, double stamp, double* out8) {
    const Vertex& v = n->vertices.at(vid);
    out8[0] = stamp; out8[1] = v.est.t[0]; out8[2] = v.est.t[1]; out8[3] = v.est.t[2];
    R_to_quat_xyzw(v.est.R, out8 + 4);
}

// Pack the active graph (edges with a non-fixed endpoint; vertices touched by such an edge) into one
// window instance.  Returns LOC_OK, or LOC_ERR_UNSUPPORTED when it does not fit the kernel's capacities.
struct Packed {
    std::vector<int32_t> counts, r_idx, p_idx, s_idx;
    std::vector<double> poses, r_val, p_val, s_val, anchors;
    std::vector<int> slot_vid;
    int band = 0;  // widest |slot_i - slot_j| over the binary edges
};

int pack(const loc_node* n, Packed& P) {
    const loc_window_caps& c = n->caps;
    std::map<int, int> slot;      // moving vertex id -> pose slot (ascending id)
    std::map<int, int> anchor_ix; // fixed vertex id -> anchor index
    auto touch = [&](int vid) {
        const Vertex& v = n->vertices.at(vid);
        if (v.fixed) { if (!anchor_ix.count(vid)) { int k = (int)anchor_ix.size(); anchor_ix[vid] = k; } }
        else slot[vid] = 0;
    };
    auto active2 = [&](int a, int b) { return !(n->vertices.at(a).fixed && n->vertices.at(b).fixed); };
    for (const auto& e : n->ranges) if (active2(e.v0, e.v1)) { touch(e.v0); touch(e.v1); }
    for (const auto& e : n->priors) if (!n->vertices.at(e.v).fixed) touch(e.v);
    for (const auto& e : n->se3s) if (active2(e.vi, e.vj)) { touch(e.vi); touch(e.vj); }
    // Pose slots in AGE order per robot (oldest first), not in vertex-id order: consecutive poses then sit next to each
    // other and the normal equations are banded (g2o orders by id; the order only changes round-off).
    int k = 0;
    P.slot_vid.clear();
    for (const RobotRing& r : n->robots) {
        if (r.is_static) continue;
        for (int i = 0; i < r.T; ++i) {
            const int vid = r.slot_vertex((r.index + 1 + i) % r.T);
            auto it = slot.find(vid);
            if (it != slot.end()) { it->second = k++; P.slot_vid.push_back(vid); }
        }
    }
    if ((int)slot.size() != k) return locamd_fail(LOC_ERR_INVALID, "internal: active vertex outside every robot ring");
    if (k > c.nv_max) return locamd_fail(LOC_ERR_UNSUPPORTED, "window has more active poses than the node was sized for");
    {
        // Oldest-first or newest-first?  Windows of more than 512 poses are factored in envelope (skyline) form in exactly
        // this order, so take the direction with the cheaper envelope: a key-frame star (addPoseEdge: every pose hangs on
        // an OLDER key pose) packed newest-first has its leaves before their key and factors without any fill, packed
        // oldest-first every row reaches back to its key.  (Smaller windows are re-ordered by the kernel anyway.)
        std::vector<int> first((size_t)k), lastn((size_t)k);
        for (int v = 0; v < k; ++v) first[(size_t)v] = lastn[(size_t)v] = v;
        auto couple = [&](int a, int b) {
            if (n->vertices.at(a).fixed || n->vertices.at(b).fixed) return;
            const int sa = slot.at(a), sb = slot.at(b), lo = std::min(sa, sb), hi = std::max(sa, sb);
            first[(size_t)hi] = std::min(first[(size_t)hi], lo);
            lastn[(size_t)lo] = std::max(lastn[(size_t)lo], hi);
        };
        for (const auto& e : n->ranges) if (active2(e.v0, e.v1)) couple(e.v0, e.v1);
        for (const auto& e : n->se3s) if (active2(e.vi, e.vj)) couple(e.vi, e.vj);
        double fwd = 0, rev = 0;
        for (int v = 0; v < k; ++v) {
            const double wf = v - first[(size_t)v] + 1, wr = lastn[(size_t)v] - v + 1;
            fwd += wf * wf; rev += wr * wr;
        }
        if (rev < fwd) {
            for (auto& kv : slot) kv.second = k - 1 - kv.second;
            std::reverse(P.slot_vid.begin(), P.slot_vid.end());
        }
    }
    P.counts.assign(4, 0);
    P.poses.assign((size_t)k * 12, 0.0);
    P.r_idx.assign(n->ranges.size() * 2, 0); P.r_val.assign(n->ranges.size() * 5, 0.0);
    P.p_idx.assign(n->priors.size(), 0); P.p_val.assign(n->priors.size() * 18, 0.0);
    P.s_idx.assign(n->se3s.size() * 4, 0); P.s_val.assign(n->se3s.size() * 48, 0.0);
    P.anchors.assign(std::max<size_t>(anchor_ix.size(), 1) * 3, 0.0);
    for (auto& kv : slot) {
        const Vertex& v = n->vertices.at(kv.first);
        std::memcpy(&P.poses[(size_t)kv.second * 12], v.est.R, sizeof(double) * 9);
        std::memcpy(&P.poses[(size_t)kv.second * 12 + 9], v.est.t, sizeof(double) * 3);
    }
    for (auto& kv : anchor_ix) std::memcpy(&P.anchors[(size_t)kv.second * 3], n->vertices.at(kv.first).est.t, sizeof(double) * 3);
    int nr = 0, np = 0, ns = 0, n_derived = 0;
    for (const auto& e : n->ranges) {
        if (!active2(e.v0, e.v1)) continue;
        int a = e.v0, b = e.v1;
        double off[3] = {e.off[0], e.off[1], e.off[2]};
        int derived_anchor = -1;
        if (n->vertices.at(a).fixed) {  // the kernel wants endpoint 0 moving; the residual is symmetric
            if (off[0] != 0 || off[1] != 0 || off[2] != 0) {
                // a lever arm on a FIXED endpoint 0 (setVertexOffset(0, ...) on a static requester, localization.cpp:334): its point
                // (X O).t = R o + t never moves — it enters as one more entry of the fixed-vertex table
                const Vertex& fv = n->vertices.at(a);
                double pt[3];
                for (int i = 0; i < 3; ++i) pt[i] = fv.est.R[3 * i] * off[0] + fv.est.R[3 * i + 1] * off[1] + fv.est.R[3 * i + 2] * off[2] + fv.est.t[i];
                if (anchor_ix.empty() && n_derived == 0) P.anchors.clear();   // (the one-entry placeholder of an anchor-less window)
                ++n_derived;
                derived_anchor = (int)(P.anchors.size() / 3);
                P.anchors.insert(P.anchors.end(), pt, pt + 3);
                off[0] = off[1] = off[2] = 0.0;
            }
            std::swap(a, b);
        }
        P.r_idx[(size_t)nr * 2] = slot.at(a);
        P.r_idx[(size_t)nr * 2 + 1] = derived_anchor >= 0 ? -1 - derived_anchor : (n->vertices.at(b).fixed ? -1 - anchor_ix.at(b) : slot.at(b));
        if (!n->vertices.at(b).fixed) P.band = std::max(P.band, std::abs(slot.at(a) - slot.at(b)));
        double* v = &P.r_val[(size_t)nr * 5];
        v[0] = e.meas; v[1] = e.info; v[2] = off[0]; v[3] = off[1]; v[4] = off[2];
        ++nr;
    }
    for (const auto& e : n->priors) {
        if (n->vertices.at(e.v).fixed) continue;
        P.p_idx[np] = slot.at(e.v);
        double* v = &P.p_val[(size_t)np * 18];
        std::memcpy(v, e.zinv.R, sizeof(double) * 9); std::memcpy(v + 9, e.zinv.t, sizeof(double) * 3);
        std::memcpy(v + 12, e.info, sizeof(double) * 6);
        ++np;
    }
    for (const auto& e : n->se3s) {
        if (!active2(e.vi, e.vj)) continue;
        if (n->vertices.at(e.vi).fixed || n->vertices.at(e.vj).fixed) return locamd_fail(LOC_ERR_UNSUPPORTED, "SE3 edge to a fixed vertex");
        int32_t* ix = &P.s_idx[(size_t)ns * 4];
        ix[0] = slot.at(e.vi); ix[1] = slot.at(e.vj); ix[2] = e.robust ? 1 : 0; ix[3] = 0;
        P.band = std::max(P.band, std::abs(ix[0] - ix[1]));
        double* v = &P.s_val[(size_t)ns * 48];
        std::memcpy(v, e.zinv.R, sizeof(double) * 9); std::memcpy(v + 9, e.zinv.t, sizeof(double) * 3);
        std::memcpy(v + 12, e.info, sizeof(double) * 36);
        ++ns;
    }
    P.counts[0] = (int32_t)slot.size(); P.counts[1] = nr; P.counts[2] = np; P.counts[3] = ns;
    P.r_idx.resize((size_t)nr * 2); P.r_val.resize((size_t)nr * 5);   // (edges between fixed vertices were skipped)
    P.p_idx.resize((size_t)np); P.p_val.resize((size_t)np * 18);
    P.s_idx.resize((size_t)ns * 4); P.s_val.resize((size_t)ns * 48);
    return LOC_OK;
}

// Capacities a solver handle needs for this packed graph, never below `have` (handles only grow: a window that slides
// keeps its shape, so after the warm-up the handle is created once).  Tight on purpose: the per-instance storage decides
// whether the window lives in LDS and how many fit a CU.
loc_window_caps grown_caps(const loc_window_caps& have, const Packed& P) {
    auto up = [](int v, int m) { return (v + m - 1) / m * m; };
    loc_window_caps c = have;
    c.nv_max = std::max(have.nv_max, std::max(1, (int)P.counts[0]));
    c.nr_max = std::max(have.nr_max, up(P.counts[1], 8));
    c.np_max = std::max(have.np_max, up(P.counts[2], 4));
    c.ns_max = std::max(have.ns_max, up(P.counts[3], 4));
    c.bw_max = std::min(c.nv_max - 1, std::max(have.bw_max, std::max(1, P.band)));
    if (c.bw_max < 0) c.bw_max = 0;
    return c;
}
bool same_caps(const loc_window_caps& a, const loc_window_caps& b) {
    return a.nv_max == b.nv_max && a.nr_max == b.nr_max && a.np_max == b.np_max && a.ns_max == b.ns_max && a.bw_max == b.bw_max;
}
// the solver reads whole [cap] rows
void pad_to(Packed& P, const loc_window_caps& c) {
    P.poses.resize((size_t)c.nv_max * 12, 0.0);
    P.r_idx.resize((size_t)c.nr_max * 2, 0); P.r_val.resize((size_t)c.nr_max * 5, 0.0);
    P.p_idx.resize((size_t)c.np_max, 0); P.p_val.resize((size_t)c.np_max * 18, 0.0);
    P.s_idx.resize((size_t)c.ns_max * 4, 0); P.s_val.resize((size_t)c.ns_max * 48, 0.0);
}

void fill_output(loc_node* n, const double* res, loc_node_output* out) {
    if (!out) return;
    out->solved = 1;
    out->chi2 = res[0];
    out->published = res[0] < n->cfg.minimum_optimize_error ? 1 : 0;  // localization.cpp:199-205
    out->outer_iterations = (int32_t)res[3];
    out->lm_trials = (int32_t)res[4];
    RobotRing* r = n->robot(n->self_id);
    pose_out(n, r->last_vertex(), r->header[r->index].stamp, out->realtime);  // current_pose(), :208
    const int idx = (r->index + 1 + n->cfg.trajectory_length / 2) % r->T;     // path->poses[T/2], :220
    pose_out(n, r->slot_vertex(idx), r->header[idx].stamp, out->optimized);
}

// Localization::solve() + publish(): one window-kernel launch on this node's graph
int solve_now(loc_node* n, loc_node_output* out) {
    Packed P;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = pack(n, P);
    if (rc != LOC_OK) return rc;
    double res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (P.counts[0] > 0) {
        // one solver handle per node, re-created only while the window is still growing into its final shape
        const loc_window_caps want = grown_caps(n->win_caps, P);
        if (n->win && !same_caps(want, n->win_caps)) { loc_window_destroy(n->win); n->win = nullptr; }
        n->win_caps = want;
        pad_to(P, want);
        if (!n->win) {
            rc = loc_window_create(&n->win, n->device, 1, &n->win_caps, (int32_t)(P.anchors.size() / 3), P.anchors.data(), n->cfg.maximum_iteration);
            if (rc != LOC_OK) return rc;
            rc = loc_window_set_jacobian(n->win, n->cfg.jacobian);
            if (rc != LOC_OK) return rc;
            // (no HIP events around the node's ~50 us kernel unless asked for at creation: they cost ~4 us per message; loc_node_last_timing's
            //  third figure is then launch-to-completion on the host clock)
            if (!getenv("LOCAMD_KERNEL_EVENTS")) (void)loc_window_set_option(n->win, "kernel_events", 0);
            n->win_anchors = P.anchors;
        } else if (n->win_anchors != P.anchors) {
            // the fixed vertices the window sees (or just their order) change as the window slides: refresh the table,
            // keep the handle — creating one costs milliseconds of hipMalloc / stream / event set-up per solve
            rc = loc_window_set_anchors(n->win, (int32_t)(P.anchors.size() / 3), P.anchors.data());
            if (rc != LOC_OK) return rc;
            n->win_anchors = P.anchors;
        }
        const auto t1 = std::chrono::steady_clock::now();
        rc = loc_window_solve_host(n->win, 1, P.counts.data(), P.poses.data(), P.r_idx.data(), P.r_val.data(), P.p_idx.data(),
                                   P.p_val.data(), P.s_idx.data(), P.s_val.data(), res);
        if (rc != LOC_OK) return rc;
        const auto t2 = std::chrono::steady_clock::now();
        n->t_pack_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        n->t_solve_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
        (void)loc_window_last_kernel_ms(n->win, &n->t_kernel_ms);
        { int32_t k = LOC_WINDOW_KERNEL_NONE; (void)loc_window_last_kernel_kind(n->win, &k); n->last_kind = k; }
        for (size_t s = 0; s < P.slot_vid.size(); ++s) {
            Vertex& v = n->vertices.at(P.slot_vid[s]);
            std::memcpy(v.est.R, &P.poses[s * 12], sizeof(double) * 9);
            std::memcpy(v.est.t, &P.poses[s * 12 + 9], sizeof(double) * 3);
        }
    }
    n->pending = false;
    fill_output(n, res, out);
    return 1;
}

int maybe_solve(loc_node* n, bool wanted, loc_node_output* out) {
    if (!wanted) return 0;
    if (n->deferred) { n->pending = true; return 0; }
    return solve_now(n, out);
}

void add_range_edge(loc_node* n, int v0, int v1, double meas, double cov, const double* off) {
    RangeEdge e{};
    e.v0 = v0; e.v1 = v1; e.meas = meas;
    e.info = 1.0 / cov;  // covariance_matrix.inverse(), localization.cpp:616-620
    if (off) { e.off[0] = off[0]; e.off[1] = off[1]; e.off[2] = off[2]; }
    n->ranges.push_back(e);
}

}  // namespace

extern "C" {

void loc_node_default_config(loc_node_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->trajectory_length = 0;            // no default in the reference (getParam, localization.cpp:72)
    c->maximum_velocity = 1.0;           // :75
    c->distance_outlier = 1.0;           // :78
    c->maximum_iteration = 20;           // :65
    c->minimum_optimize_error = 1000.0;  // :68
    c->jacobian = LOC_JAC_NUMERIC_G2O;   // what EdgeSE3Range inherits from g2o (types_edge_se3range.h:45-74); ANALYTIC is the opt-in fast mode
}

int loc_node_create(loc_node** out, int32_t device, const loc_node_config* cfg, int32_t n_nodes, const int32_t* ids,
                    const double* pos_xyz, int32_t n_antenna, const double* antenna_xyz) {
    if (!out) return locamd_fail(LOC_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!cfg || n_nodes <= 0 || !ids || !pos_xyz) return locamd_fail(LOC_ERR_INVALID, "node arguments");
    if (cfg->trajectory_length <= 0) return locamd_fail(LOC_ERR_INVALID, "robot/trajectory_length must be set");
    if (cfg->jacobian != LOC_JAC_ANALYTIC && cfg->jacobian != LOC_JAC_NUMERIC_G2O) return locamd_fail(LOC_ERR_INVALID, "jacobian mode");
    if (loc_device_count() <= 0) return locamd_fail(LOC_ERR_NO_DEVICE, "no HIP device visible: localization_amd has no CPU fallback");
    if (cfg->trajectory_length > 1024 || (cfg->has_relative_range && cfg->trajectory_length * n_nodes > 1024))
        return locamd_fail(LOC_ERR_UNSUPPORTED, "windows of more than 1024 moving poses are not supported by this kernel version");
    loc_node* n = new (std::nothrow) loc_node();
    if (!n) return locamd_fail(LOC_ERR_INVALID, "out of host memory");
    n->cfg = *cfg;
    n->device = device;
    n->self_id = ids[n_nodes - 1];  // nodesId.back(), localization.cpp:89
    {   // The only limit is the kernel's: active poses per window.  Edge capacities of the solver handle grow with what the
        // packed graphs need (grown_caps): two priors per vertex (IMU + lidar, cfg/uwb_imu_lidar.yaml), any number of
        // ranges piling onto one pose vertex (the else-branch at localization.cpp:348) — the reference has no such limits.
        const int tv = cfg->has_relative_range ? cfg->trajectory_length * n_nodes : cfg->trajectory_length;
        n->caps.nv_max = tv;
        n->caps.bw_max = -1;
        n->win_caps.nv_max = tv;                 // the window's final pose count is known; edges and band grow on demand
        n->win_caps.nr_max = 2 * tv + 8;         // the two-edge range topology (localization.cpp:327-357)
    }
    for (int i = 0; i < n_nodes; ++i) {  // :92-108 and Robot::init, robot.cpp:31-58
        RobotRing r;
        r.id = ids[i];
        r.is_static = !(cfg->has_relative_range || ids[i] == n->self_id);
        r.T = r.is_static ? 1 : cfg->trajectory_length;
        r.header.assign((size_t)r.T, Header());
        r.header[0].frame_id = "none";
        Iso p = iso_identity();
        p.t[0] = pos_xyz[i * 3]; p.t[1] = pos_xyz[i * 3 + 1]; p.t[2] = pos_xyz[i * 3 + 2];
        for (int s = 0; s < r.T; ++s) n->vertices[r.slot_vertex(s)] = Vertex{p, r.is_static};
        n->robots.push_back(r);
    }
    if (antenna_xyz && n_antenna > 0) n->antenna.assign(antenna_xyz, antenna_xyz + 3 * (size_t)n_antenna);  // :111-123
    else n->antenna.assign(9, 0.0);  // three identity offsets, localization.h:170
    *out = n;
    return LOC_OK;
}

int loc_node_destroy(loc_node* n) {
    if (n && n->win) loc_window_destroy(n->win);
    delete n;
    return LOC_OK;
}

int loc_node_set_deferred(loc_node* n, int32_t on) {
    if (!n) return locamd_fail(LOC_ERR_INVALID, "null");
    n->deferred = on != 0;
    return LOC_OK;
}
int32_t loc_node_solve_pending(const loc_node* n) { return n && n->pending ? 1 : 0; }
int loc_node_last_timing(const loc_node* n, double* pack_solve_kernel_ms) {
    if (!n || !pack_solve_kernel_ms) return locamd_fail(LOC_ERR_INVALID, "null");
    pack_solve_kernel_ms[0] = n->t_pack_ms; pack_solve_kernel_ms[1] = n->t_solve_ms; pack_solve_kernel_ms[2] = n->t_kernel_ms;
    return LOC_OK;
}

int loc_node_last_kernel_kind(const loc_node* n, int32_t* kind) {
    if (!n || !kind) return locamd_fail(LOC_ERR_INVALID, "null");
    *kind = n->last_kind;
    return LOC_OK;
}

int32_t loc_node_number_measurements(const loc_node* n) { return n ? n->number_measurements : 0; }

// Localization::addRangeEdge, localization.cpp:297-376
int loc_node_add_range(loc_node* n, int32_t requester_id, int32_t responder_id, double stamp, float distance,
                       float distance_err, int32_t antenna, const char* frame_id, loc_node_output* out) {
    if (!n) return locamd_fail(LOC_ERR_INVALID, "null");
    if (out) std::memset(out, 0, sizeof(*out));
    RobotRing* rq = n->robot(requester_id);
    RobotRing* rs = n->robot(responder_id);
    if (!rq || !rs) return locamd_fail(LOC_ERR_UNKNOWN_NODE, "node id not in nodesId");
    // (everything that can fail is checked before the graph is touched: an error return leaves the node as it was)
    if (antenna > 0 && (size_t)antenna * 3 > n->antenna.size()) return locamd_fail(LOC_ERR_INVALID, "antenna index beyond /uwb/antennaOffset");
    ++n->number_measurements;  // :303
    const Iso& xq = n->vertices.at(rq->last_vertex()).est;
    const Iso& xr = n->vertices.at(rs->last_vertex()).est;
    const double dx = xq.t[0] - xr.t[0], dy = xq.t[1] - xr.t[1], dz = xq.t[2] - xr.t[2];
    const double distance_estimation = std::sqrt(dx * dx + dy * dy + dz * dz);  // vertex origins, :306-307
    if (n->number_measurements > n->cfg.trajectory_length &&
        std::fabs(distance_estimation - (double)distance) > n->cfg.distance_outlier)
        return 0;  // rejected outlier, :309-313
    const double dt_requester = stamp - rq->header[rq->index].stamp;
    const double dt_responder = stamp - rs->header[rs->index].stamp;
    const double distance_cov = std::pow((double)distance_err, 2);                                   // :318
    const double cov_requester = std::pow(n->cfg.maximum_velocity * dt_requester / 3, 2);           // :319
    const int vertex_last_requester = rq->last_vertex();
    const int vertex_last_responder = rs->last_vertex();
    Header h; h.stamp = stamp; h.frame_id = frame_id ? frame_id : "";
    const int vertex_responder = n->new_vertex(*rs, ST_RANGE, h);  // :323
    const std::string& last_frame = rq->header[rq->index].frame_id;
    if (last_frame.find(h.frame_id) != std::string::npos || last_frame.find("none") != std::string::npos) {  // :327
        const int vertex_requester = n->new_vertex(*rq, ST_RANGE, h);
        const double* off = nullptr;
        if (antenna > 0) off = &n->antenna[(size_t)(antenna - 1) * 3];  // :333-334
        add_range_edge(n, vertex_requester, vertex_responder, (double)distance, distance_cov, off);   // :331-336
        add_range_edge(n, vertex_last_requester, vertex_requester, 0.0, cov_requester, nullptr);      // :338-340
    } else {
        add_range_edge(n, vertex_last_requester, vertex_responder, (double)distance, distance_cov + cov_requester, nullptr);  // :348-350
    }
    if (!rs->is_static) {  // :360-369
        const double cov_responder = std::pow(n->cfg.maximum_velocity * dt_responder / 3, 2);
        add_range_edge(n, vertex_last_responder, vertex_responder, 0.0, cov_responder, nullptr);
    }
    return maybe_solve(n, n->cfg.publish_range && n->number_measurements > n->cfg.trajectory_length, out);  // :371-375
}

// Localization::addRLRangeEdge, localization.cpp:378-436 (in the reference only with -DRELATIVE_LOCALIZATION, CMakeLists.txt:137)
int loc_node_add_rl_range(loc_node* n, int32_t requester_id, int32_t responder_id, double stamp, double distance,
                          const double* vel, loc_node_output* out) {
    if (!n || !vel) return locamd_fail(LOC_ERR_INVALID, "null");
    if (out) std::memset(out, 0, sizeof(*out));
    RobotRing* rq = n->robot(requester_id);
    RobotRing* rs = n->robot(responder_id);
    if (!rq || !rs) return locamd_fail(LOC_ERR_UNKNOWN_NODE, "node id not in nodesId");
    if (rq == rs) return locamd_fail(LOC_ERR_INVALID, "requester and responder are the same node");
    Header h; h.stamp = stamp; h.frame_id = "uwb";  // :380-382
    const double dt_requester = stamp - rq->header[rq->index].stamp;  // :384
    const double dt_responder = stamp - rs->header[rs->index].stamp;  // :385
    const double distance_cov = std::pow(0.054, 2);                                         // :387
    const double cov_requester = std::pow(n->cfg.maximum_velocity * dt_requester / 3, 2);   // :388
    const double cov_responder = std::pow(n->cfg.maximum_velocity * dt_responder / 3, 2);   // :389
    const int vertex_last_requester = rq->last_vertex();
    const int vertex_last_responder = rs->last_vertex();
    const int vertex_requester = n->new_vertex(*rq, ST_RANGE, h);  // :394
    const int vertex_responder = n->new_vertex(*rs, ST_RANGE, h);  // :395
    add_range_edge(n, vertex_requester, vertex_responder, distance, distance_cov, nullptr);  // :397-398
    if (!rs->is_static) add_range_edge(n, vertex_last_responder, vertex_responder, 0.0, cov_responder, nullptr);  // :400-405
    if (!rq->is_static && vertex_last_requester != vertex_requester) {  // :408-428 (T = 1 would make it a self-loop)
        Se3Edge e{};
        e.vi = vertex_last_requester; e.vj = vertex_requester; e.robust = false;   // no robust kernel is set
        Iso z = iso_identity();
        z.t[0] = dt_requester * vel[0]; z.t[1] = dt_requester * vel[1]; z.t[2] = dt_requester * vel[2];  // measurement.translate(dt v)
        e.zinv = iso_inverse(z);
        e.info[0] = e.info[7] = e.info[14] = 1.0 / cov_requester;  // translation only
        n->se3s.push_back(e);
    }
    return maybe_solve(n, n->cfg.publish_relative_range != 0, out);  // :430-434
}

// Localization::addImuEdge, localization.cpp:499-535
int loc_node_add_imu(loc_node* n, double stamp, const double* q_xyzw, const double* orientation_cov9,
                     const char* frame_id, loc_node_output* out) {
    (void)stamp;
    if (!n || !q_xyzw || !orientation_cov9 || !frame_id) return locamd_fail(LOC_ERR_INVALID, "null");
    if (out) std::memset(out, 0, sizeof(*out));
    RobotRing& r = *n->robot(n->self_id);
    if (r.header[r.index].frame_id.find(frame_id) == std::string::npos) {  // once per vertex, :501
        r.header[r.index].frame_id += std::string("-") + frame_id;         // robot.cpp:140-143
        const int vid = n->last_vertex_type(r, ST_RANGE);
        Vertex& v = n->vertices.at(vid);
        quat_to_R(q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2], v.est.R);    // rotation overwritten, translation kept, :505-513
        PriorEdge e{};
        e.v = vid;
        e.zinv = iso_inverse(v.est);
        e.info[3] = 1.0 / orientation_cov9[0]; e.info[4] = 1.0 / orientation_cov9[4]; e.info[5] = 1.0 / orientation_cov9[8];  // :516-518
        n->priors.push_back(e);
    }
    return maybe_solve(n, n->cfg.publish_imu != 0, out);
}

// Localization::addLidarEdge, localization.cpp:462-496
int loc_node_add_lidar(loc_node* n, double stamp, double z, const char* frame_id, loc_node_output* out) {
    (void)stamp;
    if (!n || !frame_id) return locamd_fail(LOC_ERR_INVALID, "null");
    if (out) std::memset(out, 0, sizeof(*out));
    RobotRing& r = *n->robot(n->self_id);
    if (r.header[r.index].frame_id.find(frame_id) == std::string::npos) {
        r.header[r.index].frame_id += std::string("-") + frame_id;
        const int vid = n->last_vertex_type(r, ST_RANGE);
        Vertex& v = n->vertices.at(vid);
        v.est.t[2] = z;  // :474
        PriorEdge e{};
        e.v = vid;
        e.zinv = iso_inverse(v.est);
        e.info[2] = 1 / 0.05;  // :479
        n->priors.push_back(e);
    }
    return maybe_solve(n, n->cfg.publish_lidar != 0, out);
}

// Localization::addPoseEdge, localization.cpp:254-290
int loc_node_add_pose(loc_node* n, double stamp, const double* pose_xyz_qxyzw, const double* cov36,
                      const char* frame_id, loc_node_output* out) {
    if (!n || !pose_xyz_qxyzw || !cov36) return locamd_fail(LOC_ERR_INVALID, "null");
    if (out) std::memset(out, 0, sizeof(*out));
    RobotRing& r = *n->robot(n->self_id);
    Header h; h.stamp = stamp; h.frame_id = frame_id ? frame_id : "";
    if (!r.hdr_has[ST_POSE]) { r.hdr_has[ST_POSE] = true; r.headers[ST_POSE] = r.header[r.index]; }  // last_header(type)
    if (h.frame_id != r.headers[ST_POSE].frame_id) n->key_vertex = n->last_vertex_type(r, ST_POSE);     // :258-259
    const int nv = n->new_vertex(r, ST_POSE, h);
    if (n->key_vertex < 0 || !n->vertices.count(n->key_vertex) || n->key_vertex == nv)
        return locamd_fail(LOC_ERR_INVALID, "key vertex left the window");  // reference: dangling pointer
    Se3Edge e{};
    e.vi = n->key_vertex; e.vj = nv; e.robust = true;
    Iso z;
    quat_to_R(pose_xyz_qxyzw[6], pose_xyz_qxyzw[3], pose_xyz_qxyzw[4], pose_xyz_qxyzw[5], z.R);  // tf::poseMsgToEigen
    z.t[0] = pose_xyz_qxyzw[0]; z.t[1] = pose_xyz_qxyzw[1]; z.t[2] = pose_xyz_qxyzw[2];
    e.zinv = iso_inverse(z);
    if (!invert6(cov36, e.info)) return locamd_fail(LOC_ERR_SINGULAR, "pose covariance is singular");  // :275-277
    n->se3s.push_back(e);
    return maybe_solve(n, n->cfg.publish_pose != 0, out);
}

// Localization::addTwistEdge + twist2transform + create_se3_edge_from_twist, localization.cpp:438-459, 560-605
int loc_node_add_twist(loc_node* n, double stamp, const double* tw, const double* cov36, const char* frame_id,
                       loc_node_output* out) {
    if (!n || !tw || !cov36) return locamd_fail(LOC_ERR_INVALID, "null");
    if (out) std::memset(out, 0, sizeof(*out));
    RobotRing& r = *n->robot(n->self_id);
    const double dt = stamp - r.header[r.index].stamp;
    const int last_vertex = r.last_vertex();
    Header h; h.stamp = stamp; h.frame_id = frame_id ? frame_id : "";
    const int nv = n->new_vertex(r, ST_TWIST, h);
    // tf::Quaternion::setRPY(wx dt, wy dt, wz dt), then tf::Matrix3x3::setRotation (normalising by 2 / |q|^2)
    const double hr = tw[3] * dt * 0.5, hp = tw[4] * dt * 0.5, hy = tw[5] * dt * 0.5;
    const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
    const double qx = sr * cp * cy - cr * sp * sy, qy = cr * sp * cy + sr * cp * sy, qz = cr * cp * sy - sr * sp * cy,
                 qw = cr * cp * cy + sr * sp * sy;
    const double s = 2.0 / (qx * qx + qy * qy + qz * qz + qw * qw);
    const double xs = qx * s, ys = qy * s, zs = qz * s;
    const double wx = qw * xs, wy = qw * ys, wz = qw * zs, xx = qx * xs, xy = qx * ys, xz = qx * zs, yy = qy * ys, yz = qy * zs, zz = qz * zs;
    Iso z;
    const double Rz[9] = {1.0 - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0 - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0 - (xx + yy)};
    std::memcpy(z.R, Rz, sizeof(Rz));
    z.t[0] = tw[0] * dt; z.t[1] = tw[1] * dt; z.t[2] = tw[2] * dt;
    Se3Edge e{};
    e.vi = last_vertex; e.vj = nv; e.robust = true;
    e.zinv = iso_inverse(z);
    double cov[36];
    for (int i = 0; i < 36; ++i) cov[i] = cov36[i] * dt * dt;  // :579
    if (!invert6(cov, e.info)) return locamd_fail(LOC_ERR_SINGULAR, "twist covariance is singular");
    if (last_vertex == nv) return locamd_fail(LOC_ERR_INVALID, "trajectory_length 1 cannot hold a twist edge");
    n->se3s.push_back(e);
    return maybe_solve(n, n->cfg.publish_twist != 0, out);
}

int loc_node_solve(loc_node* n, loc_node_output* out) {
    if (!n) return locamd_fail(LOC_ERR_INVALID, "null");
    if (out) std::memset(out, 0, sizeof(*out));
    return solve_now(n, out);
}

// Robot::vertices2path, robot.cpp:61-72: oldest -> newest, out[T][8] = stamp, xyz, q xyzw
int loc_node_get_path(loc_node* n, int32_t node_id, double* out, int32_t capacity) {
    if (!n || !out) return locamd_fail(LOC_ERR_INVALID, "null");
    RobotRing* r = n->robot(node_id);
    if (!r) return locamd_fail(LOC_ERR_UNKNOWN_NODE, "node id not in nodesId");
    if (capacity < r->T) return locamd_fail(LOC_ERR_INVALID, "path buffer too small");
    for (int i = 0; i < r->T; ++i) {
        const int idx = (r->index + 1 + i) % r->T;
        pose_out(n, r->slot_vertex(idx), r->header[idx].stamp, out + 8 * i);
    }
    return r->T;
}

// Localization::~Localization, localization.cpp:708-717: path->poses[T/2 .. T-1] of the moving tag (vertices2path order, robot.cpp:61-72)
int loc_node_flush_tail(loc_node* n, double* out, int32_t capacity) {
    if (!n || !out) return locamd_fail(LOC_ERR_INVALID, "null");
    RobotRing* r = n->robot(n->self_id);
    if (!r) return locamd_fail(LOC_ERR_UNKNOWN_NODE, "self id not in nodesId");
    const int T = n->cfg.trajectory_length, first = T / 2;
    if (capacity < T - first) return locamd_fail(LOC_ERR_INVALID, "tail buffer too small");
    for (int i = first; i < T; ++i) {
        const int idx = (r->index + 1 + i) % r->T;
        pose_out(n, r->slot_vertex(idx), r->header[idx].stamp, out + 8 * (i - first));
    }
    return T - first;
}

}  // extern "C"

// Many nodes, one launch per group: every node that has a solve pending (deferred mode) is packed into a batch with the
// other pending nodes of the same (device, maximum_iteration, jacobian) — what one kernel launch can serve.
namespace {
// cached batch solvers of the calling thread, one per group key: re-created only when the capacities or the batch size
// grow.  loc_nodes_release_batch_cache() frees them.
struct BatchCache { loc_window* w = nullptr; loc_window_caps caps{0, 0, 0, 0, 0}; int device = -1; size_t B = 0; int iters = 0; int jac = 0; };
thread_local std::vector<BatchCache> g_batch_caches;

int solve_group(loc_node** nodes, const std::vector<int>& todo, std::vector<Packed>& P, loc_node_output* outs) {
    loc_node* first = nodes[todo[0]];
    BatchCache* cache = nullptr;
    for (auto& bc : g_batch_caches)
        if (bc.device == first->device && bc.iters == first->cfg.maximum_iteration && bc.jac == first->cfg.jacobian) cache = &bc;
    if (!cache) {
        g_batch_caches.emplace_back();
        cache = &g_batch_caches.back();
        cache->device = first->device; cache->iters = first->cfg.maximum_iteration; cache->jac = first->cfg.jacobian;
    }
    loc_window_caps caps = cache->caps;
    for (int i : todo) caps = grown_caps(caps, P[(size_t)i]);
    const size_t B = todo.size();
    std::vector<int32_t> counts(B * 4), r_idx(B * caps.nr_max * 2), p_idx(B * caps.np_max), s_idx(B * caps.ns_max * 4);
    std::vector<double> poses(B * caps.nv_max * 12), r_val(B * caps.nr_max * 5), p_val(B * caps.np_max * 18),
        s_val(B * caps.ns_max * 48), res(B * 8, 0.0), anchors;
    for (size_t b = 0; b < B; ++b) {
        Packed& p = P[(size_t)todo[b]];
        const int base = (int)(anchors.size() / 3);  // every node brings its own anchors: rebase the indices
        anchors.insert(anchors.end(), p.anchors.begin(), p.anchors.end());
        for (int e = 0; e < p.counts[1]; ++e) if (p.r_idx[(size_t)e * 2 + 1] < 0) p.r_idx[(size_t)e * 2 + 1] -= base;
        std::copy(p.counts.begin(), p.counts.end(), counts.begin() + b * 4);
        std::copy(p.poses.begin(), p.poses.end(), poses.begin() + b * caps.nv_max * 12);
        std::copy(p.r_idx.begin(), p.r_idx.end(), r_idx.begin() + b * caps.nr_max * 2);
        std::copy(p.r_val.begin(), p.r_val.end(), r_val.begin() + b * caps.nr_max * 5);
        std::copy(p.p_idx.begin(), p.p_idx.end(), p_idx.begin() + b * caps.np_max);
        std::copy(p.p_val.begin(), p.p_val.end(), p_val.begin() + b * caps.np_max * 18);
        std::copy(p.s_idx.begin(), p.s_idx.end(), s_idx.begin() + b * caps.ns_max * 4);
        std::copy(p.s_val.begin(), p.s_val.end(), s_val.begin() + b * caps.ns_max * 48);
    }
    if (!cache->w || cache->B < B || !same_caps(cache->caps, caps)) {
        if (cache->w) { loc_window_destroy(cache->w); cache->w = nullptr; cache->B = 0; }
        int rc0 = loc_window_create(&cache->w, first->device, (int64_t)B, &caps, (int32_t)(anchors.size() / 3), anchors.data(), first->cfg.maximum_iteration);
        if (rc0 != LOC_OK) return rc0;
        rc0 = loc_window_set_jacobian(cache->w, first->cfg.jacobian);
        if (rc0 != LOC_OK) return rc0;
        cache->caps = caps; cache->B = B;
    } else {
        int rc0 = loc_window_set_anchors(cache->w, (int32_t)(anchors.size() / 3), anchors.data());
        if (rc0 != LOC_OK) return rc0;
    }
    int rc = loc_window_solve_host(cache->w, (int64_t)B, counts.data(), poses.data(), r_idx.data(), r_val.data(), p_idx.data(), p_val.data(),
                                   s_idx.data(), s_val.data(), res.data());
    if (rc != LOC_OK) return rc;
    for (size_t b = 0; b < B; ++b) {
        loc_node* n = nodes[todo[b]];
        const Packed& p = P[(size_t)todo[b]];
        for (size_t s = 0; s < p.slot_vid.size(); ++s) {
            Vertex& v = n->vertices.at(p.slot_vid[s]);
            std::memcpy(v.est.R, &poses[(b * caps.nv_max + s) * 12], sizeof(double) * 9);
            std::memcpy(v.est.t, &poses[(b * caps.nv_max + s) * 12 + 9], sizeof(double) * 3);
        }
        n->pending = false;
        if (outs) fill_output(n, &res[b * 8], &outs[todo[b]]);
    }
    return (int)B;
}
}  // namespace

extern "C" {

int loc_nodes_solve_batch(loc_node** nodes, int32_t n_nodes, loc_node_output* outs) {
    if (!nodes || n_nodes <= 0) return locamd_fail(LOC_ERR_INVALID, "nodes");
    std::vector<Packed> P((size_t)n_nodes);
    std::vector<int> todo;
    for (int i = 0; i < n_nodes; ++i) {
        if (!nodes[i]) return locamd_fail(LOC_ERR_INVALID, "null node");
        if (outs) std::memset(&outs[i], 0, sizeof(loc_node_output));
        if (!nodes[i]->pending) continue;
        int rc = pack(nodes[i], P[(size_t)i]);
        if (rc != LOC_OK) return rc;
        todo.push_back(i);
    }
    // a heterogeneous fleet: one launch per (device, maximum_iteration, jacobian), each node solved with ITS parameters
    int solved = 0;
    std::vector<char> done(todo.size(), 0);
    for (size_t a = 0; a < todo.size(); ++a) {
        if (done[a]) continue;
        const loc_node* na = nodes[todo[a]];
        std::vector<int> group;
        for (size_t b = a; b < todo.size(); ++b) {
            const loc_node* nb = nodes[todo[b]];
            if (!done[b] && nb->device == na->device && nb->cfg.maximum_iteration == na->cfg.maximum_iteration && nb->cfg.jacobian == na->cfg.jacobian) {
                group.push_back(todo[b]);
                done[b] = 1;
            }
        }
        const int rc = solve_group(nodes, group, P, outs);
        if (rc < 0) return rc;
        solved += rc;
    }
    return solved;
}

// Frees the calling thread's cached batch solvers (device buffers, streams, events).
int loc_nodes_release_batch_cache(void) {
    for (auto& bc : g_batch_caches) if (bc.w) loc_window_destroy(bc.w);
    g_batch_caches.clear();
    return LOC_OK;
}

}  // extern "C"
