// Host-side timing of the native ByteTrack (pa_bytetrack_update_batch) on a synthetic stream of K slowly moving boxes per
// frame, 64-frame batches (what the players tracker's host stage sees per device batch).  No GPU needed:
//     g++ -O3 -ffp-contract=off -std=c++17 -o /tmp/bt_bench tools/bytetrack_bench.cpp padel_analytics_amd/csrc/bytetrack.cpp
//     /tmp/bt_bench 95 20
// To compare against an older build of the tracker: `git show <rev>:padel_analytics_amd/csrc/bytetrack.cpp > /tmp/old.cpp` and link
// that instead (the checksum of the returned ids must be the same).  Results: profiles/r4_bytetrack_host.txt.
#include "../include/padel_hip.h"
#include <vector>
#include <random>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
int main(int argc, char** argv) {
    const int K = atoi(argv[1]), n = 64, stride = 300, reps = argc > 2 ? atoi(argv[2]) : 20;
    std::mt19937 g(1); std::normal_distribution<double> nd(0, 2); std::uniform_real_distribution<double> ux(128, 1152), uy(144, 648), us(0.5, 0.95);
    std::vector<double> cx(K), cy(K);
    for (int i = 0; i < K; ++i) { cx[i] = ux(g); cy[i] = uy(g); }
    pa_bytetrack* b; pa_bytetrack_create(0.25, 30, 0.8, 30, &b);
    std::vector<float> boxes((size_t)n * stride * 6); std::vector<int32_t> counts(n, K), ids((size_t)n * stride);
    double tot = 0; long long sum = 0;
    for (int r = 0; r < reps; ++r) {
        for (int f = 0; f < n; ++f) {
            std::vector<double> s(K); for (auto& x : s) x = us(g); std::sort(s.rbegin(), s.rend());
            for (int i = 0; i < K; ++i) {
                float* p = &boxes[((size_t)f * stride + i) * 6];
                double x = cx[i] + nd(g), y = cy[i] + nd(g);
                p[0] = x - 38; p[1] = y - 72; p[2] = x + 38; p[3] = y + 72; p[4] = s[i]; p[5] = 0;
            }
        }
        auto t0 = std::chrono::steady_clock::now();
        pa_bytetrack_update_batch(b, boxes.data(), counts.data(), nullptr, n, stride, ids.data());
        tot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int f = 0; f < n; ++f) for (int i = 0; i < K; ++i) sum += ids[(size_t)f * stride + i];
    }
    printf("K=%d: %.3f ms/batch  (%.1f us/frame) checksum %lld\n", K, 1e3 * tot / reps, 1e6 * tot / reps / n, sum);
}
