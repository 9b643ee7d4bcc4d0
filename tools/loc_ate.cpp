// loc_ate — absolute trajectory error of a TUM-format estimate against a TUM-format ground truth, the step AFTER the hot path
// (SURVEY.md §8(f-3)), as a small native tool: what the reference's manual evaluation loop computes with
//   script/associate.py:71-101     every pair of stamps with |a - (b + offset)| < max_difference is a candidate; candidates are
//                                  taken best first (ascending (difference, a, b)), each stamp is used once;
//   script/evaluate_ate.py:47-79   closed-form rigid alignment of the estimate onto the truth (Horn), :152-162 the translational
//                                  error's rmse / mean / median / std / min / max.
// Same numbers as localization_amd/ate.py (tests/test_next_rows_cpu.py compares the two on the example recording).
//
//   g++ -O2 -std=c++17 tools/loc_ate.cpp -o tools/loc_ate
//   tools/loc_ate truth.txt estimate.txt [--offset S] [--max_difference S] [--no-align]      -> one JSON line on stdout
//
// The alignment uses Horn's unit-quaternion form (the rotation is the dominant eigenvector of a symmetric 4x4 matrix built from
// the cross-covariance, found here by cyclic Jacobi rotations): the same optimum as the reference's SVD form, with the proper
// rotation (det = +1) built in.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

namespace {

struct Row { double stamp, x, y, z; };

bool read_tum(const char* path, std::vector<Row>& out) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        for (char& c : line) if (c == ',') c = ' ';
        size_t i = line.find_first_not_of(" \t\r");
        if (i == std::string::npos || line[i] == '#') continue;
        std::istringstream ss(line);
        Row r{};
        if (ss >> r.stamp >> r.x >> r.y >> r.z) out.push_back(r);
    }
    return true;
}

// associate.py:71-101 — returns (index in a, index in b), sorted by a
std::vector<std::pair<int, int>> associate(const std::vector<Row>& a, const std::vector<Row>& b, double offset, double max_difference) {
    std::vector<int> order_b(b.size());
    for (size_t j = 0; j < b.size(); ++j) order_b[j] = (int)j;
    std::stable_sort(order_b.begin(), order_b.end(), [&](int p, int q) { return b[p].stamp + offset < b[q].stamp + offset; });
    std::vector<double> bs(b.size());
    for (size_t j = 0; j < b.size(); ++j) bs[j] = b[order_b[j]].stamp + offset;
    std::vector<std::tuple<double, double, double, int, int>> cand;
    for (size_t i = 0; i < a.size(); ++i) {
        const double t = a[i].stamp;
        const size_t lo = std::lower_bound(bs.begin(), bs.end(), t - max_difference) - bs.begin();
        const size_t hi = std::upper_bound(bs.begin(), bs.end(), t + max_difference) - bs.begin();
        for (size_t jj = lo; jj < hi; ++jj) {
            const int j = order_b[jj];
            const double diff = std::fabs(t - (b[j].stamp + offset));
            if (diff < max_difference) cand.emplace_back(diff, t, b[j].stamp, (int)i, j);
        }
    }
    std::sort(cand.begin(), cand.end());
    std::vector<char> used_a(a.size(), 0), used_b(b.size(), 0);
    std::vector<std::pair<int, int>> out;
    for (const auto& c : cand) {
        const int i = std::get<3>(c), j = std::get<4>(c);
        if (used_a[i] || used_b[j]) continue;
        used_a[i] = used_b[j] = 1;
        out.emplace_back(i, j);
    }
    std::sort(out.begin(), out.end());
    return out;
}

// dominant eigenvector of a symmetric 4x4 matrix, cyclic Jacobi
void dominant_eigenvector4(double N[4][4], double q[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0;
        for (int p = 0; p < 4; ++p) for (int r = p + 1; r < 4; ++r) off += N[p][r] * N[p][r];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p)
            for (int r = p + 1; r < 4; ++r) {
                if (N[p][r] == 0.0) continue;
                const double theta = (N[r][r] - N[p][p]) / (2.0 * N[p][r]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; ++k) { const double a = N[k][p], b = N[k][r]; N[k][p] = c * a - s * b; N[k][r] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = N[p][k], b = N[r][k]; N[p][k] = c * a - s * b; N[r][k] = s * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = V[k][p], b = V[k][r]; V[k][p] = c * a - s * b; V[k][r] = s * a + c * b; }
            }
    }
    int best = 0;
    for (int k = 1; k < 4; ++k) if (N[k][k] > N[best][best]) best = k;
    for (int k = 0; k < 4; ++k) q[k] = V[k][best];
}

// rigid alignment of model onto data (evaluate_ate.py:47-79): R, t minimising sum |R m + t - d|^2
void horn_align(const std::vector<std::array<double, 3>>& model, const std::vector<std::array<double, 3>>& data, double R[3][3], double t[3]) {
    const size_t n = model.size();
    double mm[3] = {0, 0, 0}, dm[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) { mm[k] += model[i][k]; dm[k] += data[i][k]; }
    for (int k = 0; k < 3; ++k) { mm[k] /= (double)n; dm[k] /= (double)n; }
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // sum (m - mm)(d - dm)^T
    for (size_t i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) S[r][c] += (model[i][r] - mm[r]) * (data[i][c] - dm[c]);
    double N[4][4] = {
        {S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0]},
        {S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2]},
        {S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1]},
        {S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]}};
    double q[4];
    dominant_eigenvector4(N, q);
    const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double w = q[0] / nq, x = q[1] / nq, y = q[2] / nq, z = q[3] / nq;
    R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z); R[0][2] = 2 * (x * z + w * y);
    R[1][0] = 2 * (x * y + w * z); R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
    R[2][0] = 2 * (x * z - w * y); R[2][1] = 2 * (y * z + w * x); R[2][2] = 1 - 2 * (x * x + y * y);
    for (int r = 0; r < 3; ++r) t[r] = dm[r] - (R[r][0] * mm[0] + R[r][1] * mm[1] + R[r][2] * mm[2]);
}

}  // namespace

int main(int argc, char** argv) {
    const char* truth_path = nullptr;
    const char* est_path = nullptr;
    double offset = 0.0, max_difference = 0.02;
    bool align = true;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--offset") && i + 1 < argc) offset = std::atof(argv[++i]);
        else if (!std::strcmp(argv[i], "--max_difference") && i + 1 < argc) max_difference = std::atof(argv[++i]);
        else if (!std::strcmp(argv[i], "--no-align")) align = false;
        else if (!truth_path) truth_path = argv[i];
        else if (!est_path) est_path = argv[i];
        else { std::fprintf(stderr, "loc_ate: unexpected argument %s\n", argv[i]); return 2; }
    }
    if (!truth_path || !est_path) {
        std::fprintf(stderr, "usage: loc_ate truth.txt estimate.txt [--offset S] [--max_difference S] [--no-align]\n");
        return 2;
    }
    std::vector<Row> truth, est;
    if (!read_tum(truth_path, truth)) { std::fprintf(stderr, "loc_ate: cannot read %s\n", truth_path); return 1; }
    if (!read_tum(est_path, est)) { std::fprintf(stderr, "loc_ate: cannot read %s\n", est_path); return 1; }
    const auto pairs = associate(truth, est, offset, max_difference);
    if (pairs.size() < 2) {
        std::fprintf(stderr, "Couldn't find matching timestamp pairs between groundtruth and estimated trajectory\n");   // evaluate_ate.py:139
        return 1;
    }
    std::vector<std::array<double, 3>> a, b;   // truth, estimate
    for (const auto& p : pairs) {
        a.push_back({truth[p.first].x, truth[p.first].y, truth[p.first].z});
        b.push_back({est[p.second].x, est[p.second].y, est[p.second].z});
    }
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
    if (align) horn_align(b, a, R, t);
    std::vector<double> err(a.size());
    double s2 = 0, s1 = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        double e2 = 0;
        for (int r = 0; r < 3; ++r) {
            const double v = R[r][0] * b[i][0] + R[r][1] * b[i][1] + R[r][2] * b[i][2] + t[r] - a[i][r];
            e2 += v * v;
        }
        err[i] = std::sqrt(e2);
        s2 += e2; s1 += err[i];
    }
    const double n = (double)err.size(), mean = s1 / n;
    double var = 0;
    for (double e : err) var += (e - mean) * (e - mean);
    std::vector<double> sorted = err;
    std::sort(sorted.begin(), sorted.end());
    const double median = sorted.size() % 2 ? sorted[sorted.size() / 2] : 0.5 * (sorted[sorted.size() / 2 - 1] + sorted[sorted.size() / 2]);
    std::printf("{\"pairs\": %zu, \"rmse\": %.12g, \"mean\": %.12g, \"median\": %.12g, \"std\": %.12g, \"min\": %.12g, \"max\": %.12g}\n",
                err.size(), std::sqrt(s2 / n), mean, median, std::sqrt(var / n), sorted.front(), sorted.back());
    return 0;
}
