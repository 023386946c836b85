// Host-native ByteTrack (pa_bytetrack_*): the stateful, frame-sequential association step that follows the
// players detector (reference: `self.byte_track.update_with_detections(detections)` at
// trackers/players_tracker/players_tracker.py:367-369, constructed at :311 with frame_rate = fps).
//
// Same algorithm, same orderings and the same id semantics as padel_analytics_amd/bytetrack.py (the documented
// Python restatement of supervision's ByteTrack, pinned by tests/golden/bytetrack_golden.json); this file exists
// because the engine returns ~10^2 boxes per frame for 64-frame batches every ~25 ms and per-frame Python cannot
// keep up with that.  One call consumes a whole batch of frames in order and returns a track id per box (-1 =
// dropped).  The assignment solver is the shortest-augmenting-path algorithm of Crouse (2016) in the exact
// iteration order scipy.optimize.linear_sum_assignment uses, so both implementations pick the same optimum when
// several exist.  Pure host code: no HIP calls, usable without a GPU.
#include "../../include/padel_hip.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

namespace {

enum { ST_NEW = 0, ST_TRACKED = 1, ST_LOST = 2, ST_REMOVED = 3 };
constexpr double WP = 1.0 / 20, WV = 1.0 / 160;

struct Trk {
    double tlwh[4]{};
    double score = 0;
    double mean[8]{};
    double cov[64]{};
    bool has_mean = false, activated = false;
    int state = ST_NEW;
    long long internal_id = 0, track_id = -1;
    int frame_id = 0, start_frame = 0, tracklet_len = 0;
};
using P = std::shared_ptr<Trk>;

void tlbr_of(const Trk& t, double* r) {
    if (t.has_mean) {
        r[0] = t.mean[0]; r[1] = t.mean[1]; r[2] = t.mean[2]; r[3] = t.mean[3];
        r[2] *= r[3];
        r[0] -= r[2] / 2; r[1] -= r[3] / 2;
    } else {
        memcpy(r, t.tlwh, sizeof(double) * 4);
    }
    r[2] += r[0]; r[3] += r[1];
}

void to_xyah(const double* tlwh, double* r) {
    r[0] = tlwh[0] + tlwh[2] / 2; r[1] = tlwh[1] + tlwh[3] / 2; r[2] = tlwh[2] / tlwh[3]; r[3] = tlwh[3];
}

// cost[i][j] = 1 - IoU(a_i, b_j), row-major na x nb
std::vector<double> iou_distance(const std::vector<double>& a, int na, const std::vector<double>& b, int nb) {
    std::vector<double> c((size_t)na * nb);
    for (int i = 0; i < na; ++i) {
        const double* p = &a[i * 4];
        const double area_a = (p[2] - p[0]) * (p[3] - p[1]);
        for (int j = 0; j < nb; ++j) {
            const double* q = &b[j * 4];
            const double area_b = (q[2] - q[0]) * (q[3] - q[1]);
            const double w = std::max(std::min(p[2], q[2]) - std::max(p[0], q[0]), 0.0);
            const double h = std::max(std::min(p[3], q[3]) - std::max(p[1], q[1]), 0.0);
            const double inter = w * h;
            c[(size_t)i * nb + j] = 1.0 - inter / (area_a + area_b - inter);
        }
    }
    return c;
}

std::vector<double> boxes_of(const std::vector<P>& ts) {
    std::vector<double> b(ts.size() * 4);
    for (size_t i = 0; i < ts.size(); ++i) tlbr_of(*ts[i], &b[i * 4]);
    return b;
}

// ---- rectangular linear sum assignment (Crouse 2016; iteration order of scipy's rectangular_lsap)
long augmenting_path(long nc, const double* cost, std::vector<double>& u, std::vector<double>& v, std::vector<long>& path,
                     std::vector<long>& row4col, std::vector<double>& spc, long i, std::vector<char>& SR,
                     std::vector<char>& SC, std::vector<long>& remaining, double* p_min) {
    double min_val = 0;
    long num_remaining = nc;
    for (long it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    std::fill(SR.begin(), SR.end(), 0);
    std::fill(SC.begin(), SC.end(), 0);
    std::fill(spc.begin(), spc.end(), INFINITY);
    long sink = -1;
    while (sink == -1) {
        long index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (long it = 0; it < num_remaining; ++it) {
            const long j = remaining[it];
            const double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        const long j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

// -> pairs (row, col) sorted by row
bool lsa(long nr, long nc, const std::vector<double>& cost_in, std::vector<std::pair<long, long>>& out) {
    out.clear();
    if (nr == 0 || nc == 0) return true;
    const bool transpose = nc < nr;
    std::vector<double> tmp;
    const double* cost = cost_in.data();
    if (transpose) {
        tmp.resize((size_t)nr * nc);
        for (long i = 0; i < nr; ++i)
            for (long j = 0; j < nc; ++j) tmp[(size_t)j * nr + i] = cost_in[(size_t)i * nc + j];
        std::swap(nr, nc);
        cost = tmp.data();
    }
    std::vector<double> u(nr, 0), v(nc, 0), spc(nc);
    std::vector<long> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (long cur = 0; cur < nr; ++cur) {
        double min_val;
        const long sink = augmenting_path(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
        if (sink < 0) return false;
        u[cur] += min_val;
        for (long i = 0; i < nr; ++i) if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (long j = 0; j < nc; ++j) if (SC[j]) v[j] -= min_val - spc[j];
        long j = sink;
        while (true) {
            const long i = path[j];
            row4col[j] = i;
            std::swap(col4row[i], j);
            if (i == cur) break;
        }
    }
    if (transpose) {
        std::vector<long> idx(nr);
        for (long i = 0; i < nr; ++i) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](long a, long b) { return col4row[a] < col4row[b]; });
        for (long k : idx) out.emplace_back(col4row[k], k);
    } else {
        for (long i = 0; i < nr; ++i) out.emplace_back(i, col4row[i]);
    }
    return true;
}

struct Assign { std::vector<std::pair<int, int>> matches; std::vector<int> u_rows, u_cols; };

Assign linear_assignment(std::vector<double> cost, int nr, int nc, double thresh) {
    Assign a;
    if ((size_t)nr * nc == 0) {
        for (int i = 0; i < nr; ++i) a.u_rows.push_back(i);
        for (int j = 0; j < nc; ++j) a.u_cols.push_back(j);
        return a;
    }
    for (double& c : cost) if (c > thresh) c = thresh + 1e-4;
    std::vector<std::pair<long, long>> sol;
    lsa(nr, nc, cost, sol);
    std::vector<char> mr(nr, 0), mc(nc, 0);
    for (auto& rc : sol)
        if (cost[(size_t)rc.first * nc + rc.second] <= thresh) {
            a.matches.emplace_back((int)rc.first, (int)rc.second);
            mr[rc.first] = 1; mc[rc.second] = 1;
        }
    for (int i = 0; i < nr; ++i) if (!mr[i]) a.u_rows.push_back(i);
    for (int j = 0; j < nc; ++j) if (!mc[j]) a.u_cols.push_back(j);
    return a;
}

// ---- Kalman filter on (cx, cy, aspect, h, and their velocities)
void kf_initiate(Trk& t) {
    double m[4];
    to_xyah(t.tlwh, m);
    for (int i = 0; i < 4; ++i) { t.mean[i] = m[i]; t.mean[4 + i] = 0; }
    const double h = m[3];
    const double sd[8] = {2 * WP * h, 2 * WP * h, 1e-2, 2 * WP * h, 10 * WV * h, 10 * WV * h, 1e-5, 10 * WV * h};
    std::fill(t.cov, t.cov + 64, 0.0);
    for (int i = 0; i < 8; ++i) t.cov[i * 9] = sd[i] * sd[i];
    t.has_mean = true;
}

void kf_predict(Trk& t) {
    if (t.state != ST_TRACKED) t.mean[7] = 0;
    const double h = t.mean[3];
    const double sd[8] = {WP * h, WP * h, 1e-2, WP * h, WV * h, WV * h, 1e-5, WV * h};
    for (int i = 0; i < 4; ++i) t.mean[i] += t.mean[4 + i];
    // P' = F P F^T + Q with F = I + shift(4): rows / columns i < 4 gain row / column i + 4
    double fp[64];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) fp[i * 8 + j] = t.cov[i * 8 + j] + (i < 4 ? t.cov[(i + 4) * 8 + j] : 0.0);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) t.cov[i * 8 + j] = fp[i * 8 + j] + (j < 4 ? fp[i * 8 + j + 4] : 0.0);
    for (int i = 0; i < 8; ++i) t.cov[i * 9] += sd[i] * sd[i];
}

void kf_update(Trk& t, const double* meas) {
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
    for (int c = 3; c >= 0; --c) {
        for (int k = 0; k < 8; ++k) {
            double s = Bt[c][k];
            for (int r = c + 1; r < 4; ++r) s -= Sc[c * 4 + r] * Bt[r][k];
            Bt[c][k] = s / Sc[c * 4 + c];
        }
    }
    double innov[4];
    for (int i = 0; i < 4; ++i) innov[i] = meas[i] - t.mean[i];
    for (int r = 0; r < 8; ++r) {
        double s = 0;
        for (int i = 0; i < 4; ++i) s += Bt[i][r] * innov[i];
        t.mean[r] += s;
    }
    // P -= K S K^T
    double KS[8][4];
    for (int r = 0; r < 8; ++r)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int i = 0; i < 4; ++i) s += Bt[i][r] * S[i * 4 + j];
            KS[r][j] = s;
        }
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            double s = 0;
            for (int j = 0; j < 4; ++j) s += KS[r][j] * Bt[j][c];
            t.cov[r * 8 + c] -= s;
        }
}

std::vector<P> joint(const std::vector<P>& a, const std::vector<P>& b) {
    std::vector<P> out = a;
    for (const P& t : b) {
        bool seen = false;
        for (const P& s : a) if (s->internal_id == t->internal_id) { seen = true; break; }
        if (!seen) out.push_back(t);
    }
    return out;
}

std::vector<P> sub(const std::vector<P>& a, const std::vector<P>& b) {
    std::vector<P> out;
    for (const P& t : a) {
        bool in_b = false;
        for (const P& s : b) if (s->internal_id == t->internal_id) { in_b = true; break; }
        if (!in_b) out.push_back(t);
    }
    return out;
}

}  // namespace

struct pa_bytetrack {
    double track_thresh, match_thresh, det_thresh;
    int max_time_lost;
    int frame_id = 0;
    long long next_internal = 0, next_id = 0;
    std::vector<P> tracked, lost;

    void confirm(Trk& t) {
        t.activated = true;
        if (t.track_id == -1) t.track_id = ++next_id;
    }

    // one frame: boxes n x 4 (x1,y1,x2,y2), scores n -> ids[n] (-1: dropped)
    void update(const double* boxes, const double* scores, int n, int32_t* ids) {
        const int fid = ++frame_id;
        std::vector<P> activated, refind, lost_now, removed;
        auto mk = [&](int i) {
            P t = std::make_shared<Trk>();
            t->tlwh[0] = boxes[i * 4]; t->tlwh[1] = boxes[i * 4 + 1];
            t->tlwh[2] = boxes[i * 4 + 2] - boxes[i * 4]; t->tlwh[3] = boxes[i * 4 + 3] - boxes[i * 4 + 1];
            t->score = scores[i];
            return t;
        };
        std::vector<P> dets, dets2;
        for (int i = 0; i < n; ++i) {
            if (scores[i] > track_thresh) dets.push_back(mk(i));
            else if (scores[i] > 0.1 && scores[i] < track_thresh) dets2.push_back(mk(i));
        }
        std::vector<P> unconfirmed, trk;
        for (const P& t : tracked) (t->activated ? trk : unconfirmed).push_back(t);
        std::vector<P> pool = joint(trk, lost);
        for (const P& t : pool) kf_predict(*t);

        auto apply = [&](const Assign& a, const std::vector<P>& tr, const std::vector<P>& de) {
            for (auto& m : a.matches) {
                Trk& t = *tr[m.first];
                double meas[4];
                to_xyah(de[m.second]->tlwh, meas);
                kf_update(t, meas);
                t.score = de[m.second]->score;
            }
            for (auto& m : a.matches) {
                const P& t = tr[m.first];
                if (t->state == ST_TRACKED) { t->tracklet_len += 1; activated.push_back(t); }
                else { t->tracklet_len = 0; refind.push_back(t); }
                t->state = ST_TRACKED;
                t->frame_id = fid;
                confirm(*t);
            }
        };
        auto fused = [&](const std::vector<P>& tr, const std::vector<P>& de) {
            std::vector<double> c = iou_distance(boxes_of(tr), (int)tr.size(), boxes_of(de), (int)de.size());
            for (size_t i = 0; i < tr.size(); ++i)
                for (size_t j = 0; j < de.size(); ++j) {
                    double& x = c[i * de.size() + j];
                    x = 1.0 - (1.0 - x) * de[j]->score;
                }
            return c;
        };

        Assign a1 = linear_assignment(fused(pool, dets), (int)pool.size(), (int)dets.size(), match_thresh);
        apply(a1, pool, dets);
        std::vector<P> r_tracked;
        for (int i : a1.u_rows) if (pool[i]->state == ST_TRACKED) r_tracked.push_back(pool[i]);
        Assign a2 = linear_assignment(iou_distance(boxes_of(r_tracked), (int)r_tracked.size(), boxes_of(dets2), (int)dets2.size()),
                                      (int)r_tracked.size(), (int)dets2.size(), 0.5);
        apply(a2, r_tracked, dets2);
        for (int i : a2.u_rows) {
            const P& t = r_tracked[i];
            if (t->state != ST_LOST) { t->state = ST_LOST; lost_now.push_back(t); }
        }
        std::vector<P> rest;
        for (int j : a1.u_cols) rest.push_back(dets[j]);
        Assign a3 = linear_assignment(fused(unconfirmed, rest), (int)unconfirmed.size(), (int)rest.size(), 0.7);
        apply(a3, unconfirmed, rest);
        for (int i : a3.u_rows) { unconfirmed[i]->state = ST_REMOVED; removed.push_back(unconfirmed[i]); }
        for (int j : a3.u_cols) {
            const P& t = rest[j];
            if (t->score < det_thresh) continue;
            t->internal_id = ++next_internal;
            kf_initiate(*t);
            t->tracklet_len = 0;
            t->state = ST_TRACKED;
            t->frame_id = t->start_frame = fid;
            if (fid == 1) confirm(*t);
            activated.push_back(t);
        }
        for (const P& t : lost)
            if (fid - t->frame_id > max_time_lost) { t->state = ST_REMOVED; removed.push_back(t); }
        std::vector<P> keep;
        for (const P& t : tracked) if (t->state == ST_TRACKED) keep.push_back(t);
        tracked = joint(joint(keep, activated), refind);
        std::vector<P> l = sub(lost, tracked);
        l.insert(l.end(), lost_now.begin(), lost_now.end());
        l = sub(l, removed);
        lost.clear();
        for (const P& t : l) if (t->state == ST_LOST) lost.push_back(t);
        // remove duplicates between tracked and lost (IoU distance < 0.15: keep the longer-lived one)
        {
            const int na = (int)tracked.size(), nb = (int)lost.size();
            std::vector<double> d = iou_distance(boxes_of(tracked), na, boxes_of(lost), nb);
            std::vector<char> da(na, 0), db(nb, 0);
            for (int p = 0; p < na; ++p)
                for (int q = 0; q < nb; ++q)
                    if (d[(size_t)p * nb + q] < 0.15) {
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
        for (const P& t : tracked) if (t->activated) out.push_back(t);
        for (int i = 0; i < n; ++i) ids[i] = -1;
        if (!out.empty() && n > 0) {
            std::vector<double> db(boxes, boxes + (size_t)n * 4);
            Assign a = linear_assignment(iou_distance(db, n, boxes_of(out), (int)out.size()), n, (int)out.size(), 0.5);
            for (auto& m : a.matches) ids[m.first] = (int32_t)out[m.second]->track_id;
        }
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
