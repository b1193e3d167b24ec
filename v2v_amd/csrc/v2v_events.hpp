// v2v_events.hpp -- event-list -> voxel-grid scatter kernels (gfx950).
//
// Replaces the NumPy scatter voxelisers of the reference:
//   data/testh5.py:60-90            TestH5Dataset.make_voxel  (== scripts/visualize_esim_sample.py:113-135)
//   utils/event_utils.py:692-728    events_to_voxel (temporal_bilinear=True) + events_to_image :155-174
// One work-item per event; float64 global atomics (global_atomic_add_f64) into the [Tb,H,W] grid.  Discrete
// bins add +-1 (order-independent, exact).  Interpolated bins add float64 weights whose per-event values are
// bitwise NumPy's (same IEEE expression); only the summation ORDER differs from np.add.at / np.bincount, so
// sums agree to ~1e-15 relative (tested at 1e-12), far inside the 1e-5 bar.
// Bound: atomic throughput (8 B per event and touched bin), not HBM streaming.
// Index semantics follow the reference's scatter primitive: np.add.at (make_voxel) and index_put_ (the torch twin) WRAP an
// index in [-size, -1] once, np.ravel_multi_index (events_to_voxel) raises on it; anything else outside the sensor raises in
// all three.  Events that would raise are dropped and counted in `dropped` (the Python wrapper raises IndexError when it is
// non-zero); events of the segmented form that lie in no interval are ignored, wherever they point.
// A zero time span (one event, or equal timestamps) makes the bilinear weights 0/0: the reference propagates NaN into every
// bin of the touched pixels (np.maximum / torch.max keep NaN); so do these kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2v {

enum { kEvMakeVoxelDiscrete = 0, kEvMakeVoxelInterp = 1, kEvBilinear = 2 };

struct EventArgs {
    const int64_t *seg;      // optional [F+1] ascending event offsets: events [seg[f], seg[f+1]) form voxel grid f (nullptr: one grid)
    int64_t n_seg;
    const double *ts;
    const int64_t *xs, *ys;
    const double *ps;
    int64_t n;
    int32_t mode, Tb;
    int64_t H, W;
    double *out;
    unsigned long long *dropped;
};

__global__ void __launch_bounds__(256) events_to_voxel_kernel(const EventArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    // segmented form (TestH5Dataset.__getitem__, data/testh5.py:111-119: one make_voxel per image interval): the
    // interval of event i is found by binary search; its first / last timestamps play the role of ts[0] / ts[-1]
    int64_t lo = 0, hi = a.n, f = 0;
    if (a.seg) {
        int64_t l = 0, r = a.n_seg;                                      // seg[l] <= i < seg[r]
        while (r - l > 1) { const int64_t m = (l + r) >> 1; if (a.seg[m] <= i) l = m; else r = m; }
        f = l; lo = a.seg[l]; hi = a.seg[l + 1];
        if (i < lo || i >= hi) return;                                   // event outside every interval: not this call's business
    }
    int64_t x = a.xs[i], y = a.ys[i];
    if (a.mode != kEvBilinear) {                                         // np.add.at wraps negative indices once
        if (x < 0) x += a.W;
        if (y < 0) y += a.H;
    }
    if (x < 0 || x >= a.W || y < 0 || y >= a.H) { atomicAdd(a.dropped, 1ull); return; }
    const double t0 = a.ts[lo], t1 = a.ts[hi - 1], t = a.ts[i];
    const int64_t plane = a.H * a.W;
    double *cell = a.out + f * a.Tb * plane + y * a.W + x;
    if (a.mode == kEvBilinear) {
        // event_utils.py:713-719: dt = ts[-1]-ts[0]; t_norm = (ts-ts[0])/dt*(B-1); w_b = max(0, 1-|t_norm-b|)
        const double dt = t1 - t0;
        const double t_norm = (t - t0) / dt * (double)(a.Tb - 1);
        const double p = a.ps[i];
        if (!(t_norm == t_norm)) {                                       // dt == 0 -> 0/0: NaN weights in every bin, like NumPy
            for (int b = 0; b < a.Tb; ++b) atomicAdd(cell + (int64_t)b * plane, t_norm * p);
            return;
        }
        int b0 = (int)floor(t_norm);
        for (int b = (b0 < 0 ? 0 : b0); b <= b0 + 1 && b < a.Tb; ++b) {
            const double w = 1.0 - fabs(t_norm - (double)b);
            if (w > 0.0) atomicAdd(cell + (int64_t)b * plane, p * w);
        }
        return;
    }
    // make_voxel (testh5.py:67-82): ps {0,1} -> {-1,+1}; ts -> int64 microseconds since the first event
    const double pol = (double)((int)(int8_t)a.ps[i] * 2 - 1);
    const int64_t tus = (int64_t)((t - t0) * 1e6);
    const int64_t tus_last = (int64_t)((t1 - t0) * 1e6);
    if (a.mode == kEvMakeVoxelDiscrete) {
        const double t_per_bin = ((double)tus_last + 0.001) / (double)a.Tb;
        const uint8_t b = (uint8_t)(int64_t)floor((double)tus / t_per_bin);     // .astype(np.uint8)
        if (b < a.Tb) atomicAdd(cell + (int64_t)b * plane, pol);
        else atomicAdd(a.dropped, 1ull);
    } else {
        const double dt = (double)(tus_last - 0);
        const double t_norm = (double)(tus - 0) / (dt + 0.0001) * (double)(a.Tb - 1);
        const int b0 = (int)floor(t_norm);
        for (int b = (b0 < 0 ? 0 : b0); b <= b0 + 1 && b < a.Tb; ++b) {
            const double w = 1.0 - fabs(t_norm - (double)b);
            if (w > 0.0) atomicAdd(cell + (int64_t)b * plane, w * pol);
        }
    }
}

// ---- float32 twin: events_to_voxel_torch (utils/event_utils.py:466-507) + events_to_image_torch (:330-376) --------
// Per-event terms are computed in float32 exactly as torch does on the CPU (IEEE division/multiply); the
// accumulation is float32 atomics, i.e. index_put_(accumulate=True) with a different (and, as in torch on a GPU,
// unspecified) summation order.
struct EventArgsF32 {
    const float *ts;          // plain form: float32 timestamps as the caller's torch tensor holds them
    const double *ts64;       // segmented form: float64 timestamps; (ts - ts[first of the interval]).astype(float32) is applied here
    const int64_t *seg;       // optional [F+1] ascending event offsets (with ts64): interval f -> grid f
    int64_t n_seg;
    int32_t min_events;       // intervals with fewer events stay zero (data/dataset.py:189-190: `if len(xs) < 3: empty voxel`)
    const int64_t *xs, *ys;
    const float *ps;
    int64_t n;
    int32_t discrete, Tb;
    int64_t H, W;
    float *out;
    unsigned long long *dropped;
};

__global__ void __launch_bounds__(256) events_to_voxel_f32_kernel(const EventArgsF32 a)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    int64_t lo = 0, hi = a.n, f = 0;
    if (a.seg) {
        int64_t l = 0, r = a.n_seg;
        while (r - l > 1) { const int64_t m = (l + r) >> 1; if (a.seg[m] <= i) l = m; else r = m; }
        f = l; lo = a.seg[l]; hi = a.seg[l + 1];
        if (i < lo || i >= hi || hi - lo < a.min_events) return;
    }
    int64_t x = a.xs[i], y = a.ys[i];
    if (x < 0) x += a.W;                                     // index_put_ wraps negative indices once
    if (y < 0) y += a.H;
    if (x < 0 || x >= a.W || y < 0 || y >= a.H) { atomicAdd(a.dropped, 1ull); return; }
    float t0, dt, t;
    if (a.ts64) {                                            // data/dataset.py:194: ts = (ts - ts_0).astype(np.float32), ts[0] == 0
        t0 = 0.0f;
        t = (float)(a.ts64[i] - a.ts64[lo]);
        dt = (float)(a.ts64[hi - 1] - a.ts64[lo]) - t0;
    } else {
        t0 = a.ts[lo]; t = a.ts[i]; dt = a.ts[hi - 1] - t0;
    }
    const float p = a.ps[i];
    const int64_t plane = a.H * a.W;
    float *cell = a.out + f * a.Tb * plane + y * a.W + x;
    if (a.discrete) {                                        // :502-505
        const float t_per_bin = (dt + 0.001f) / (float)a.Tb;
        const int b = (int)floorf((t - t0) / t_per_bin);
        if (b >= 0 && b < a.Tb) atomicAdd(cell + (int64_t)b * plane, p);
        else atomicAdd(a.dropped, 1ull);
        return;
    }
    const float t_norm = (t - t0) / dt * (float)(a.Tb - 1);  // :491
    if (!(t_norm == t_norm)) {                               // dt == 0: NaN weights in every bin, like torch
        for (int b = 0; b < a.Tb; ++b) atomicAdd(cell + (int64_t)b * plane, p * t_norm);
        return;
    }
    const int b0 = (int)floorf(t_norm);
    for (int b = (b0 < 0 ? 0 : b0); b <= b0 + 1 && b < a.Tb; ++b) {
        const float w = 1.0f - fabsf(t_norm - (float)b);     // :495-496
        if (w > 0.0f) atomicAdd(cell + (int64_t)b * plane, p * w);
    }
}

}  // namespace v2v
