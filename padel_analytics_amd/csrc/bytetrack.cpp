// Host-native ByteTrack (pa_bytetrack_*): the stateful, frame-sequential association step that follows the
// players detector (reference: `self.byte_track.update_with_detections(detections)` at
// trackers/players_tracker/players_tracker.py:367-369, constructed at :311 with frame_rate = fps).
//
// Same algorithm, same orderings and the same id semantics as padel_analytics_amd/bytetrack.py (the documented
// Python restatement of supervision's ByteTrack, pinned by tests/golden/bytetrack_golden.json); this file exists
// because the engine returns ~10^2 boxes per frame for 64-frame batches every ~10 ms and per-frame Python cannot
// keep up with that.  One call consumes a whole batch of frames in order and returns a track id per box (-1 =
// dropped).  The assignment solver is the shortest-augmenting-path algorithm of Crouse (2016) in the exact
// iteration order scipy.optimize.linear_sum_assignment uses, so both implementations pick the same optimum when
// several exist.  Pure host code: no HIP calls, usable without a GPU.
//
// Round 4: the host stage of the players tracker had become the bound of the runner on the 640-pixel workloads (12.6 ms per
// 64-frame batch at ~95 boxes per frame against 11 ms / 5.5 ms of device time, f32 / f16).  Same arithmetic — every double is
// produced by the same operations in the same order as before, no contraction (-ffp-contract=off), ids bit-identical
// (tests/test_bytetrack_golden.py: goldens, dense random streams against the Python twin) — reorganised:
//   * the solver's per-row setup (five array fills) folded into a first pass that writes what it used to initialise, visited
//     rows / columns kept as lists, the pick under scipy's tie rule found by a second, branch-light pass;
//   * cost matrices built in place from structure-of-arrays boxes (inner loops over contiguous doubles: they vectorise),
//     clipped where they are built, no copies into the solver;
//   * the Kalman update's loops turned so that the contiguous index is innermost (sums keep their term order);
//   * no Trk object per detection (only new tracks allocate), track lists as plain pointers with a sweep per frame,
//     set operations by stamp instead of id searches.
#include "../../include/padel_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

namespace {

#define PA_HOT __attribute__((target_clones("avx2", "default")))

enum { ST_NEW = 0, ST_TRACKED = 1, ST_LOST = 2, ST_REMOVED = 3 };
constexpr double WP = 1.0 / 20, WV = 1.0 / 160;

struct Trk {
    double tlwh[4]{};
    double score = 0;
    double mean[8]{};
    double cov[64]{};
    bool activated = false;
    int state = ST_NEW;
    long long internal_id = 0, track_id = -1;
    int frame_id = 0, start_frame = 0, tracklet_len = 0;
    unsigned mark = 0, live = 0;
};
using P = Trk*;

// boxes as structure of arrays (x1, y1, x2, y2, area)
struct Boxes {
    std::vector<double> x1, y1, x2, y2, area;
    int n = 0;
    void resize(int k) { n = k; x1.resize(k); y1.resize(k); x2.resize(k); y2.resize(k); area.resize(k); }
    void set(int i, double a, double b, double c, double d) { x1[i] = a; y1[i] = b; x2[i] = c; y2[i] = d; area[i] = (c - a) * (d - b); }
};

// tlbr of a track's Kalman mean (every track in a list has one)
void set_track_box(Boxes& b, int i, const Trk& t) {
    double r[4] = {t.mean[0], t.mean[1], t.mean[2], t.mean[3]};
    r[2] *= r[3];
    r[0] -= r[2] / 2; r[1] -= r[3] / 2;
    r[2] += r[0]; r[3] += r[1];
    b.set(i, r[0], r[1], r[2], r[3]);
}
void boxes_of(const std::vector<P>& ts, Boxes& b) {
    b.resize((int)ts.size());
    for (size_t i = 0; i < ts.size(); ++i) set_track_box(b, (int)i, *ts[i]);
}

void to_xyah(const double* tlwh, double* r) {
    r[0] = tlwh[0] + tlwh[2] / 2; r[1] = tlwh[1] + tlwh[3] / 2; r[2] = tlwh[2] / tlwh[3]; r[3] = tlwh[3];
}

// cost[i][j] = 1 - IoU(a_i, b_j), row-major na x nb; with `score`: 1 - IoU * score_j ("fuse_score"); with clip > 0: entries
// above `clip` become clip + 1e-4 (what linear_assignment does to its copy before solving)
PA_HOT void iou_cost(const Boxes& a, const Boxes& b, const double* score, double clip, double* c) {
    const int na = a.n, nb = b.n;
    const double* bx1 = b.x1.data(); const double* by1 = b.y1.data(); const double* bx2 = b.x2.data(); const double* by2 = b.y2.data();
    const double* barea = b.area.data();
    const double over = clip + 1e-4;
    for (int i = 0; i < na; ++i) {
        const double p0 = a.x1[i], p1 = a.y1[i], p2 = a.x2[i], p3 = a.y2[i], area_a = a.area[i];
        double* ci = c + (size_t)i * nb;
        for (int j = 0; j < nb; ++j) {
            const double w = std::max(std::min(p2, bx2[j]) - std::max(p0, bx1[j]), 0.0);
            const double h = std::max(std::min(p3, by2[j]) - std::max(p1, by1[j]), 0.0);
            const double inter = w * h;
            double x = 1.0 - inter / (area_a + barea[j] - inter);
            if (score) x = 1.0 - (1.0 - x) * score[j];
            if (clip > 0 && x > clip) x = over;
            ci[j] = x;
        }
    }
}

// ---- rectangular linear sum assignment (Crouse 2016; iteration order of scipy's rectangular_lsap)
struct Lsa {
    std::vector<double> u, v, spc, tcost;
    std::vector<int> path, col4row, row4col, remaining, srows, scols, idx;

    // shortest augmenting path from row `cur`; visited rows / columns are left in srows / scols
    PA_HOT int augment(int nc, const double* cost, int cur, int* n_sr, int* n_sc, double* p_min) {
        double* __restrict spc_ = spc.data();
        const double* __restrict v_ = v.data();
        const int* __restrict row4col_ = row4col.data();
        int* __restrict path_ = path.data();
        int nsr = 0, nsc = 0;
        int i = cur;
        double min_val = 0;
        int num_remaining = nc;
        int sink = -1;
        // first iteration: every column is remaining (remaining[it] = nc - 1 - it) and spc is +inf everywhere, so `r < spc[j]`
        // is `r < inf` and the pass writes all of spc / path instead of initialising them
        {
            srows[nsr++] = i;
            const double* __restrict ci = cost + (size_t)i * nc;
            const double ui = u[i];
            for (int j = 0; j < nc; ++j) {
                const double r = min_val + ci[j] - ui - v_[j];
                spc_[j] = r < INFINITY ? r : INFINITY;
            }
            for (int j = 0; j < nc; ++j) path_[j] = i;   // (a column whose r is not < inf keeps spc = inf and is never entered)
            // the minimum over four interleaved partial minima (no NaN in spc; which zero of -0 / +0 comes out cannot matter:
            // it is only compared and added)
            double m0 = INFINITY, m1 = INFINITY, m2 = INFINITY, m3 = INFINITY;
            int j4 = 0;
            for (; j4 + 4 <= nc; j4 += 4) {
                m0 = spc_[j4] < m0 ? spc_[j4] : m0;
                m1 = spc_[j4 + 1] < m1 ? spc_[j4 + 1] : m1;
                m2 = spc_[j4 + 2] < m2 ? spc_[j4 + 2] : m2;
                m3 = spc_[j4 + 3] < m3 ? spc_[j4 + 3] : m3;
            }
            for (; j4 < nc; ++j4) m0 = spc_[j4] < m0 ? spc_[j4] : m0;
            m0 = m1 < m0 ? m1 : m0;
            m2 = m3 < m2 ? m3 : m2;
            const double lowest = m2 < m0 ? m2 : m0;
            min_val = lowest;
            if (min_val == INFINITY) return -1;
            // the pick: scanning it = 0 .. nc - 1 (j = nc - 1 .. 0), the first position at the minimum wins, then every later
            // position at the minimum whose column is unassigned replaces it  ->  the smallest unassigned j at the minimum if
            // there is one, else the largest j at the minimum
            int jsel = -1;
            for (int j = 0; j < nc; ++j)
                if (spc_[j] == lowest && row4col_[j] == -1) { jsel = j; break; }
            if (jsel < 0)
                for (int j = nc - 1; j >= 0; --j)
                    if (spc_[j] == lowest) { jsel = j; break; }
            const int j = jsel;
            scols[nsc++] = j;
            if (row4col_[j] == -1) {
                sink = j;
            } else {
                i = row4col_[j];
                // the general loop needs `remaining` as the reference algorithm would have it now
                for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
                remaining[nc - 1 - j] = remaining[--num_remaining];
            }
        }
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            srows[nsr++] = i;
            const double* ci = cost + (size_t)i * nc;
            const double ui = u[i];
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + ci[j] - ui - v_[j];
                if (r < spc_[j]) { path_[j] = i; spc_[j] = r; }
                if (spc_[j] < lowest || (spc_[j] == lowest && row4col_[j] == -1)) { lowest = spc_[j]; index = it; }
            }
            min_val = lowest;
            if (min_val == INFINITY) return -1;
            const int j = remaining[index];
            if (row4col_[j] == -1) sink = j; else i = row4col_[j];
            scols[nsc++] = j;
            remaining[index] = remaining[--num_remaining];
        }
        *n_sr = nsr; *n_sc = nsc; *p_min = min_val;
        return sink;
    }

    // cost: row-major nr x nc -> pairs (row, col) sorted by row
    bool solve(int nr, int nc, const double* cost_in, std::vector<std::pair<int, int>>& out) {
        out.clear();
        if (nr == 0 || nc == 0) return true;
        const bool transpose = nc < nr;
        const double* cost = cost_in;
        if (transpose) {
            tcost.resize((size_t)nr * nc);
            for (int i = 0; i < nr; ++i)
                for (int j = 0; j < nc; ++j) tcost[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
            std::swap(nr, nc);
            cost = tcost.data();
        }
        u.assign(nr, 0.0); v.assign(nc, 0.0); spc.resize(nc);
        path.assign(nc, -1); col4row.assign(nr, -1); row4col.assign(nc, -1); remaining.resize(nc);
        srows.resize(nr + 1); scols.resize(nc + 1);
        for (int cur = 0; cur < nr; ++cur) {
            double min_val;
            int nsr, nsc;
            const int sink = augment(nc, cost, cur, &nsr, &nsc, &min_val);
            if (sink < 0) return false;
            u[cur] += min_val;
            for (int k = 0; k < nsr; ++k) { const int i = srows[k]; if (i != cur) u[i] += min_val - spc[col4row[i]]; }
            for (int k = 0; k < nsc; ++k) { const int j = scols[k]; v[j] -= min_val - spc[j]; }
            int j = sink;
            while (true) {
                const int i = path[j];
                row4col[j] = i;
                std::swap(col4row[i], j);
                if (i == cur) break;
            }
        }
        if (transpose) {
            idx.resize(nr);
            for (int i = 0; i < nr; ++i) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return col4row[a] < col4row[b]; });
            for (int k : idx) out.emplace_back(col4row[k], k);
        } else {
            for (int i = 0; i < nr; ++i) out.emplace_back(i, col4row[i]);
        }
        return true;
    }
};

struct Assign {
    std::vector<std::pair<int, int>> matches;
    std::vector<int> u_rows, u_cols;
    void clear() { matches.clear(); u_rows.clear(); u_cols.clear(); }
};

// ---- Kalman filter on (cx, cy, aspect, h, and their velocities)
void kf_initiate(Trk& t) {
    double m[4];
    to_xyah(t.tlwh, m);
    for (int i = 0; i < 4; ++i) { t.mean[i] = m[i]; t.mean[4 + i] = 0; }
    const double h = m[3];
    const double sd[8] = {2 * WP * h, 2 * WP * h, 1e-2, 2 * WP * h, 10 * WV * h, 10 * WV * h, 1e-5, 10 * WV * h};
    std::fill(t.cov, t.cov + 64, 0.0);
    for (int i = 0; i < 8; ++i) t.cov[i * 9] = sd[i] * sd[i];
}

PA_HOT void kf_predict(Trk& t) {
    if (t.state != ST_TRACKED) t.mean[7] = 0;
    const double h = t.mean[3];
    const double sd[8] = {WP * h, WP * h, 1e-2, WP * h, WV * h, WV * h, 1e-5, WV * h};
    for (int i = 0; i < 4; ++i) t.mean[i] += t.mean[4 + i];
    // P' = F P F^T + Q with F = I + shift(4): rows / columns i < 4 gain row / column i + 4
    double fp[64];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) fp[i * 8 + j] = t.cov[i * 8 + j] + t.cov[(i + 4) * 8 + j];
    for (int i = 4; i < 8; ++i)
        for (int j = 0; j < 8; ++j) fp[i * 8 + j] = t.cov[i * 8 + j] + 0.0;
    for (int i = 0; i < 8; ++i) {
        for (int j = 0; j < 4; ++j) t.cov[i * 8 + j] = fp[i * 8 + j] + fp[i * 8 + j + 4];
        for (int j = 4; j < 8; ++j) t.cov[i * 8 + j] = fp[i * 8 + j] + 0.0;
    }
    for (int i = 0; i < 8; ++i) t.cov[i * 9] += sd[i] * sd[i];
}

PA_HOT void kf_update(Trk& t, const double* meas) {
    const double h = t.mean[3];
    const double sd[4] = {WP * h, WP * h, 1e-1, WP * h};
    double S[16], Bt[4][8];                     // S = H P H^T + R ; Bt = (P H^T)^T, solved in place -> K^T
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = t.cov[i * 8 + j] + (i == j ? sd[i] * sd[i] : 0.0);
    double Sc[16];
    memcpy(Sc, S, sizeof(S));
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 8; ++r) Bt[i][r] = t.cov[r * 8 + i];
    // Gaussian elimination with partial pivoting: S X = Bt  (X = K^T, 4 x 8)
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(Sc[r * 4 + c]) > std::fabs(Sc[piv * 4 + c])) piv = r;
        if (piv != c) {
            for (int k = 0; k < 4; ++k) std::swap(Sc[c * 4 + k], Sc[piv * 4 + k]);
            for (int k = 0; k < 8; ++k) std::swap(Bt[c][k], Bt[piv][k]);
        }
        for (int r = c + 1; r < 4; ++r) {
            const double f = Sc[r * 4 + c] / Sc[c * 4 + c];
            for (int k = c; k < 4; ++k) Sc[r * 4 + k] -= f * Sc[c * 4 + k];
            for (int k = 0; k < 8; ++k) Bt[r][k] -= f * Bt[c][k];
        }
    }
    // back substitution; per column k the terms are subtracted for r = c + 1 .. 3 in that order
    for (int c = 3; c >= 0; --c) {
        for (int r = c + 1; r < 4; ++r) {
            const double f = Sc[c * 4 + r];
            for (int k = 0; k < 8; ++k) Bt[c][k] -= f * Bt[r][k];
        }
        const double d = Sc[c * 4 + c];
        for (int k = 0; k < 8; ++k) Bt[c][k] = Bt[c][k] / d;
    }
    double innov[4];
    for (int i = 0; i < 4; ++i) innov[i] = meas[i] - t.mean[i];
    // sums over i = 0 .. 3 in that order, starting from 0 (0 + x is x)
    double dm[8];
    for (int r = 0; r < 8; ++r) dm[r] = 0.0 + Bt[0][r] * innov[0];
    for (int i = 1; i < 4; ++i)
        for (int r = 0; r < 8; ++r) dm[r] += Bt[i][r] * innov[i];
    for (int r = 0; r < 8; ++r) t.mean[r] += dm[r];
    // P -= K S K^T
    double KSt[4][8];                           // KSt[j][r] = (K S)[r][j]
    for (int j = 0; j < 4; ++j) {
        for (int r = 0; r < 8; ++r) KSt[j][r] = 0.0 + Bt[0][r] * S[0 * 4 + j];
        for (int i = 1; i < 4; ++i)
            for (int r = 0; r < 8; ++r) KSt[j][r] += Bt[i][r] * S[i * 4 + j];
    }
    for (int r = 0; r < 8; ++r) {
        double s[8];
        for (int c = 0; c < 8; ++c) s[c] = 0.0 + KSt[0][r] * Bt[0][c];
        for (int j = 1; j < 4; ++j)
            for (int c = 0; c < 8; ++c) s[c] += KSt[j][r] * Bt[j][c];
        for (int c = 0; c < 8; ++c) t.cov[r * 8 + c] -= s[c];
    }
}

}  // namespace

struct pa_bytetrack {
    double track_thresh, match_thresh, det_thresh;
    int max_time_lost;
    int frame_id = 0;
    long long next_internal = 0, next_id = 0;
    std::vector<P> tracked, lost;
    std::vector<std::unique_ptr<Trk>> owned;          // every track alive: freed by the sweep at the end of a frame
    unsigned stamp = 0;

    // scratch kept across frames
    Lsa lsa;
    Boxes ba, bb;
    std::vector<double> cost, dscore;
    Assign a1, a2, a3, a4;
    std::vector<std::pair<int, int>> sol;
    std::vector<char> mr, mc;

    void confirm(Trk& t) {
        t.activated = true;
        if (t.track_id == -1) t.track_id = ++next_id;
    }

    // a ++ [t in b : t not in a]   (identity = internal id = the object for everything that has been initiated)
    void joint(const std::vector<P>& a, const std::vector<P>& b, std::vector<P>& out) {
        ++stamp;
        out = a;
        for (P t : a) t->mark = stamp;
        for (P t : b) if (t->mark != stamp) out.push_back(t);
    }
    // [t in a : t not in b]
    void sub(const std::vector<P>& a, const std::vector<P>& b, std::vector<P>& out) {
        ++stamp;
        out.clear();
        for (P t : b) t->mark = stamp;
        for (P t : a) if (t->mark != stamp) out.push_back(t);
    }

    // `cost` (nr x nc, already clipped at thresh) -> matches at cost <= thresh + unmatched rows / columns
    void assign(int nr, int nc, double thresh, Assign& a) {
        a.clear();
        if ((size_t)nr * nc == 0) {
            for (int i = 0; i < nr; ++i) a.u_rows.push_back(i);
            for (int j = 0; j < nc; ++j) a.u_cols.push_back(j);
            return;
        }
        lsa.solve(nr, nc, cost.data(), sol);
        mr.assign(nr, 0); mc.assign(nc, 0);
        for (auto& rc : sol)
            if (cost[(size_t)rc.first * nc + rc.second] <= thresh) {
                a.matches.emplace_back(rc.first, rc.second);
                mr[rc.first] = 1; mc[rc.second] = 1;
            }
        for (int i = 0; i < nr; ++i) if (!mr[i]) a.u_rows.push_back(i);
        for (int j = 0; j < nc; ++j) if (!mc[j]) a.u_cols.push_back(j);
    }

    // one frame: boxes n x 4 (x1,y1,x2,y2), scores n -> ids[n] (-1: dropped)
    void update(const double* boxes, const double* scores, int n, int32_t* ids) {
        const int fid = ++frame_id;
        std::vector<P> activated, refind, lost_now, removed;
        std::vector<int> dets, dets2;                   // detection indices: first / second association
        for (int i = 0; i < n; ++i) {
            if (scores[i] > track_thresh) dets.push_back(i);
            else if (scores[i] > 0.1 && scores[i] < track_thresh) dets2.push_back(i);
        }
        auto det_tlwh = [&](int d, double* r) {
            r[0] = boxes[d * 4]; r[1] = boxes[d * 4 + 1]; r[2] = boxes[d * 4 + 2] - boxes[d * 4]; r[3] = boxes[d * 4 + 3] - boxes[d * 4 + 1];
        };
        // the box a detection is matched by: tlwh -> tlbr as the reference's STrack property computes it
        auto det_boxes = [&](const std::vector<int>& de, Boxes& b) {
            b.resize((int)de.size());
            dscore.resize(de.size());
            for (size_t k = 0; k < de.size(); ++k) {
                double r[4];
                det_tlwh(de[k], r);
                r[2] += r[0]; r[3] += r[1];
                b.set((int)k, r[0], r[1], r[2], r[3]);
                dscore[k] = scores[de[k]];
            }
        };
        std::vector<P> unconfirmed, trk, pool;
        for (P t : tracked) (t->activated ? trk : unconfirmed).push_back(t);
        joint(trk, lost, pool);
        for (P t : pool) kf_predict(*t);

        auto apply = [&](const Assign& a, const std::vector<P>& tr, const std::vector<int>& de) {
            for (auto& m : a.matches) {
                Trk& t = *tr[m.first];
                double tlwh[4], meas[4];
                det_tlwh(de[m.second], tlwh);
                to_xyah(tlwh, meas);
                kf_update(t, meas);
                t.score = scores[de[m.second]];
            }
            for (auto& m : a.matches) {
                P t = tr[m.first];
                if (t->state == ST_TRACKED) { t->tracklet_len += 1; activated.push_back(t); }
                else { t->tracklet_len = 0; refind.push_back(t); }
                t->state = ST_TRACKED;
                t->frame_id = fid;
                confirm(*t);
            }
        };
        auto build = [&](const std::vector<P>& tr, const std::vector<int>& de, bool fuse, double thresh) {
            boxes_of(tr, ba);
            det_boxes(de, bb);
            cost.resize(tr.size() * de.size());
            if (!tr.empty() && !de.empty()) iou_cost(ba, bb, fuse ? dscore.data() : nullptr, thresh, cost.data());
        };

        build(pool, dets, true, match_thresh);
        assign((int)pool.size(), (int)dets.size(), match_thresh, a1);
        apply(a1, pool, dets);
        std::vector<P> r_tracked;
        for (int i : a1.u_rows) if (pool[i]->state == ST_TRACKED) r_tracked.push_back(pool[i]);
        build(r_tracked, dets2, false, 0.5);
        assign((int)r_tracked.size(), (int)dets2.size(), 0.5, a2);
        apply(a2, r_tracked, dets2);
        for (int i : a2.u_rows) {
            P t = r_tracked[i];
            if (t->state != ST_LOST) { t->state = ST_LOST; lost_now.push_back(t); }
        }
        std::vector<int> rest;
        for (int j : a1.u_cols) rest.push_back(dets[j]);
        build(unconfirmed, rest, true, 0.7);
        assign((int)unconfirmed.size(), (int)rest.size(), 0.7, a3);
        apply(a3, unconfirmed, rest);
        for (int i : a3.u_rows) { unconfirmed[i]->state = ST_REMOVED; removed.push_back(unconfirmed[i]); }
        for (int j : a3.u_cols) {
            const int d = rest[j];
            if (scores[d] < det_thresh) continue;
            owned.emplace_back(new Trk());
            P t = owned.back().get();
            det_tlwh(d, t->tlwh);
            t->score = scores[d];
            t->internal_id = ++next_internal;
            kf_initiate(*t);
            t->tracklet_len = 0;
            t->state = ST_TRACKED;
            t->frame_id = t->start_frame = fid;
            if (fid == 1) confirm(*t);
            activated.push_back(t);
        }
        for (P t : lost)
            if (fid - t->frame_id > max_time_lost) { t->state = ST_REMOVED; removed.push_back(t); }
        std::vector<P> keep, tmp, l;
        for (P t : tracked) if (t->state == ST_TRACKED) keep.push_back(t);
        joint(keep, activated, tmp);
        joint(tmp, refind, tracked);
        sub(lost, tracked, l);
        l.insert(l.end(), lost_now.begin(), lost_now.end());
        sub(l, removed, tmp);
        lost.clear();
        for (P t : tmp) if (t->state == ST_LOST) lost.push_back(t);
        // remove duplicates between tracked and lost (IoU distance < 0.15: keep the longer-lived one)
        if (!tracked.empty() && !lost.empty()) {
            const int na = (int)tracked.size(), nb = (int)lost.size();
            boxes_of(tracked, ba);
            boxes_of(lost, bb);
            cost.resize((size_t)na * nb);
            iou_cost(ba, bb, nullptr, 0.0, cost.data());
            std::vector<char> da(na, 0), db(nb, 0);
            for (int p = 0; p < na; ++p)
                for (int q = 0; q < nb; ++q)
                    if (cost[(size_t)p * nb + q] < 0.15) {
                        if (tracked[p]->frame_id - tracked[p]->start_frame > lost[q]->frame_id - lost[q]->start_frame) db[q] = 1;
                        else da[p] = 1;
                    }
            std::vector<P> ta, lb;
            for (int p = 0; p < na; ++p) if (!da[p]) ta.push_back(tracked[p]);
            for (int q = 0; q < nb; ++q) if (!db[q]) lb.push_back(lost[q]);
            tracked.swap(ta);
            lost.swap(lb);
        }
        // detections <- ids of the active tracks they overlap (update_with_detections)
        std::vector<P> out;
        for (P t : tracked) if (t->activated) out.push_back(t);
        for (int i = 0; i < n; ++i) ids[i] = -1;
        if (!out.empty() && n > 0) {
            ba.resize(n);
            for (int i = 0; i < n; ++i) ba.set(i, boxes[i * 4], boxes[i * 4 + 1], boxes[i * 4 + 2], boxes[i * 4 + 3]);
            boxes_of(out, bb);
            cost.resize((size_t)n * out.size());
            iou_cost(ba, bb, nullptr, 0.5, cost.data());
            assign(n, (int)out.size(), 0.5, a4);
            for (auto& m : a4.matches) ids[m.first] = (int32_t)out[m.second]->track_id;
        }
        // free what no list holds any more
        ++stamp;
        for (P t : tracked) t->live = stamp;
        for (P t : lost) t->live = stamp;
        owned.erase(std::remove_if(owned.begin(), owned.end(), [&](const std::unique_ptr<Trk>& t) { return t->live != stamp; }), owned.end());
    }
};

extern "C" {

int pa_bytetrack_create(double track_activation_threshold, int lost_track_buffer, double minimum_matching_threshold,
                        int frame_rate, pa_bytetrack** out) {
    if (!out) return 1;
    pa_bytetrack* b = new pa_bytetrack();
    b->track_thresh = track_activation_threshold;
    b->match_thresh = minimum_matching_threshold;
    b->det_thresh = track_activation_threshold + 0.1;
    b->max_time_lost = (int)((double)frame_rate / 30.0 * lost_track_buffer);
    *out = b;
    return 0;
}

void pa_bytetrack_destroy(pa_bytetrack* b) { delete b; }

void pa_bytetrack_reset(pa_bytetrack* b) {
    if (!b) return;
    b->frame_id = 0;
    b->next_internal = b->next_id = 0;
    b->tracked.clear();
    b->lost.clear();
    b->owned.clear();
}

int pa_bytetrack_update_batch(pa_bytetrack* b, const float* boxes, const int32_t* counts, const uint8_t* keep,
                              int n_frames, int stride, int32_t* out_ids) {
    if (!b || !boxes || !counts || !out_ids || n_frames < 0 || stride < 1) return 1;
    std::vector<double> bx, sc;
    std::vector<int> src;
    std::vector<int32_t> ids;
    for (int f = 0; f < n_frames; ++f) {
        const int n = std::min(counts[f], stride);
        bx.clear(); sc.clear(); src.clear();
        for (int i = 0; i < n; ++i) {
            out_ids[(size_t)f * stride + i] = -1;
            if (keep && !keep[(size_t)f * stride + i]) continue;
            const float* r = boxes + ((size_t)f * stride + i) * 6;
            bx.push_back(r[0]); bx.push_back(r[1]); bx.push_back(r[2]); bx.push_back(r[3]);
            sc.push_back(r[4]);
            src.push_back(i);
        }
        for (int i = n; i < stride; ++i) out_ids[(size_t)f * stride + i] = -1;
        ids.assign(src.size(), -1);
        b->update(bx.data(), sc.data(), (int)src.size(), ids.data());
        for (size_t k = 0; k < src.size(); ++k) out_ids[(size_t)f * stride + src[k]] = ids[k];
    }
    return 0;
}

}  // extern "C"
