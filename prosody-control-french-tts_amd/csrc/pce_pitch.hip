// pce_pitch.hip -- Praat autocorrelation pitch (R1/R2) on gfx950.
//
// Replaces parselmouth Sound.to_pitch(pitch_floor, pitch_ceiling) + selected_array +
// voiced median / geometric mean (Code/audioPipeline.py:326-335,
// Code/Pipeline/compute_pitch_adjustments.py:167-208), i.e. Praat's
// Sound_to_Pitch_ac (AC_HANNING) and Pitch_pathFinder.  The published algorithm is
// restated in oracle/pce_oracle.c; this file is its MI355X execution plan:
//
//   k_energy (shared)   per-slice integer sum / min / max  -> global mean and peak
//   k_pitch_frames      one WAVEFRONT per analysis frame.  The frame (window of
//                       3/floor s, <= ~1.8k samples) is staged once in LDS as
//                       mean-subtracted, Hann-windowed fp64; lanes own autocorrelation
//                       lags (direct sums out of LDS, the broadcast operand is shared by
//                       the wave), the normalised autocorrelation r[-L..L] stays in LDS,
//                       local maxima are found with a ballot, and every sin(x)/x
//                       interpolation (depth 30 / 70 / 700) spreads its terms over the 64
//                       lanes and finishes with a butterfly reduction.  Candidates live
//                       one-per-lane in registers.  Brent's minimiser runs wave-uniform.
//   k_pitch_path        one wavefront per slice: Viterbi over <= 16 candidates per frame
//                       (lane = current candidate x 4-way split of the previous ones),
//                       candidates staged through LDS in 64-frame tiles, back-pointers in
//                       global memory, back-tracking through LDS tiles.
//   k_pitch_median      one workgroup per slice: bitonic sort of the voiced F0 in LDS.
//
// Roofline: k_pitch_frames is fp64-VALU / LDS bound (about 10^3 flop per algorithmic
// byte), not HBM bound: algorithmic traffic is 2 B of PCM per
// sample plus 8 B of F0 per frame.
#include "pce_internal.h"
#include <cmath>

int pce_energy_plan(pce_ctx *c, const pce_slice *slices, int32_t n, DevBuf &work_buf, DevBuf &out_buf, int64_t *n_work);
int pce_energy_launch(pce_ctx *c, int32_t n, int32_t loud_thr, int64_t n_work, DevBuf &work_buf, DevBuf &out_buf);
void pce_energy_range_ptrs(const DevBuf &out_buf, size_t *stride_bytes, const long long **sum, const int **m_hi, const int **m_lo);

// ---------------------------------------------------------------------------
// host: the sizes Sound_to_Pitch_any derives before its frame loop
// ---------------------------------------------------------------------------
int pitch_plan_make(int64_t nx, double dx, double x1, const pce_pitch_params *p, PitchPlan *pl)
{
    double dt = p->time_step, minimumPitch = p->pitch_floor, periodsPerWindow = p->periods_per_window;
    double ceiling = p->pitch_ceiling;
    int64_t maxnCandidates = p->max_candidates;
    if (nx < 1 || !(dx > 0.0) || !(minimumPitch > 0.0) || !(periodsPerWindow > 0.0)) return PCE_SLICE_TOO_SHORT;
    if (maxnCandidates < 2) maxnCandidates = 2;
    if ((double)maxnCandidates < ceiling / minimumPitch) maxnCandidates = (int64_t)std::floor(ceiling / minimumPitch);
    if (dt <= 0.0) dt = periodsPerWindow / minimumPitch / 4.0;
    const double duration = dx * (double)nx;
    if (minimumPitch < periodsPerWindow / duration) return PCE_SLICE_TOO_SHORT;
    pl->nsamp_period = (int64_t)std::floor(1.0 / dx / minimumPitch);
    pl->halfnsamp_period = pl->nsamp_period / 2 + 1;
    if (ceiling > 0.5 / dx) ceiling = 0.5 / dx;
    pl->dt_window = periodsPerWindow / minimumPitch;
    pl->nsamp_window = (int64_t)std::floor(pl->dt_window / dx);
    pl->halfnsamp_window = pl->nsamp_window / 2 - 1;
    if (pl->halfnsamp_window < 2) return PCE_SLICE_TOO_SHORT;
    pl->nsamp_window = pl->halfnsamp_window * 2;
    pl->maximum_lag = (int64_t)std::floor((double)pl->nsamp_window / periodsPerWindow) + 2;
    if (pl->maximum_lag > pl->nsamp_window) pl->maximum_lag = pl->nsamp_window;
    const double myDuration = dx * (double)nx;
    if (pl->dt_window > myDuration) return PCE_SLICE_TOO_SHORT;
    pl->n_frames = (int64_t)std::floor((myDuration - pl->dt_window) / dt) + 1;
    if (pl->n_frames < 1) return PCE_SLICE_TOO_SHORT;
    const double ourMidTime = x1 - 0.5 * dx + 0.5 * myDuration;
    const double thyDuration = (double)pl->n_frames * dt;
    pl->t1 = ourMidTime - 0.5 * thyDuration + 0.5 * dt;
    pl->dt = dt; pl->ceiling = ceiling; pl->max_candidates = maxnCandidates;
    pl->brent_ixmax = (int64_t)((double)pl->nsamp_window * 0.5);
    return PCE_SLICE_OK;
}

namespace {

constexpr double PI_D = 3.1415926535897932384626433832795028841972;
constexpr double LOG2E_D = 1.4426950408889634073599246810018921374266;
constexpr int PI_WPB = 4;                 // waves (= frames) per block in k_pitch_frames
constexpr int PI_MAXC = 16;               // candidates per frame the kernels can hold
constexpr int PATH_TILE = 64;             // frames staged per LDS tile in k_pitch_path

struct PiParams {
    double dx, dt, min_pitch, ceiling, voicing_thr, octave_cost, silence_thr, oj_cost, vuv_cost;
    int nsp, hsp, nw, hw, maxlag, bix, maxc, xs_len, rr_len, pad;
};
struct PiSlice {
    int64_t begin, clip_len, clip_off, nx, frame_off;
    double x1, t1;
    int32_t n_frames, status;
};
struct PiWork { int32_t slice, frame0; };

__device__ __forceinline__ double wave_sum_f64(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Praat NUM_interpolate_sinc with its terms spread over the wave.  y is the LDS image of
// r[-bix..bix] (0-based storage of Praat's 1-based y[1..nx]); every lane passes the same x
// and receives the same result.
__device__ double sinc_wave(const double *y, int nx, double x, int maxDepth, int lane)
{
    const int midleft = (int)floor(x), midright = midleft + 1;
    if (x > (double)nx) return y[nx - 1];
    if (x < 1.0) return y[0];
    if (x == (double)midleft) return y[midleft - 1];
    if (maxDepth > midright - 1) maxDepth = midright - 1;
    if (maxDepth > nx - midleft) maxDepth = nx - midleft;
    if (maxDepth <= 0) return y[(int)floor(x + 0.5) - 1];
    if (maxDepth == 1) return y[midleft - 1] + (x - (double)midleft) * (y[midright - 1] - y[midleft - 1]);
    if (maxDepth == 2) {
        const double yl = y[midleft - 1], yr = y[midright - 1];
        const double dyl = 0.5 * (yr - y[midleft - 2]), dyr = 0.5 * (y[midright] - yl);
        const double fil = x - (double)midleft, fir = (double)midright - x;
        return yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2.0 * (yr - yl)));
    }
    const int left = midright - maxDepth, right = midleft + maxDepth;
    const double a_l = PI_D * (x - (double)midleft), a_r = PI_D * ((double)midright - x);
    const double hs_l = 0.5 * sin(a_l), hs_r = 0.5 * sin(a_r);
    const double den_l = x - (double)left + 1.0, den_r = (double)right - x + 1.0;
    const double aa_l = a_l / den_l, daa_l = PI_D / den_l;
    const double aa_r = a_r / den_r, daa_r = PI_D / den_r;
    double acc = 0.0;
    for (int t = lane; t < 2 * maxDepth; t += 64) {
        const bool is_left = t < maxDepth;
        const int k = is_left ? t : t - maxDepth;
        const double kd = (double)k;
        const double a = (is_left ? a_l : a_r) + kd * PI_D;
        const double aa = (is_left ? aa_l : aa_r) + kd * (is_left ? daa_l : daa_r);
        double hs = is_left ? hs_l : hs_r;
        if (k & 1) hs = -hs;
        const int ix = is_left ? midleft - k : midright + k;          // 1-based
        const double d = hs / a * (1.0 + cos(aa));
        acc += y[ix - 1] * d;
    }
    return wave_sum_f64(acc);
}

// Praat NUMimproveExtremum (maximum) = NUMminimize_brent on -sinc, tol 1e-10, <= 60 iterations.
__device__ double improve_maximum_wave(const double *y, int nx, int ixmid, int depth, double *ixmid_real, int lane)
{
    if (ixmid <= 1) { *ixmid_real = 1.0; return y[0]; }
    if (ixmid >= nx) { *ixmid_real = (double)nx; return y[nx - 1]; }
    double a = (double)(ixmid - 1), b = (double)(ixmid + 1);
    const double golden = 1.0 - 0.6180339887498948482045868343656381177203;
    const double sqrt_epsilon = 1.4901161193847656e-08;     // sqrt(DBL_EPSILON)
    const double tol = 1e-10;
    double v = a + golden * (b - a);
    double fv = -sinc_wave(y, nx, v, depth, lane);
    double x = v, w = v, fx = fv, fw = fv;
    for (int iter = 1; iter <= 60; iter++) {
        const double range = b - a;
        const double middle_range = (a + b) / 2.0;
        const double tol_act = sqrt_epsilon * fabs(x) + tol / 3.0;
        if (fabs(x - middle_range) + range / 2.0 <= 2.0 * tol_act) break;
        double new_step = golden * (x < middle_range ? b - x : a - x);
        if (fabs(x - w) >= tol_act) {
            double t = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * t;
            q = 2.0 * (q - t);
            if (q > 0.0) p = -p; else q = -q;
            if (fabs(p) < fabs(new_step * q) && p > q * (a - x + 2.0 * tol_act) && p < q * (b - x - 2.0 * tol_act))
                new_step = p / q;
        }
        if (fabs(new_step) < tol_act) new_step = new_step > 0.0 ? tol_act : -tol_act;
        const double t = x + new_step;
        const double ft = -sinc_wave(y, nx, t, depth, lane);
        if (ft <= fx) {
            if (t < x) b = x; else a = x;
            v = w; w = x; x = t;
            fv = fw; fw = fx; fx = ft;
        } else {
            if (t < x) a = t; else b = t;
            if (ft <= fw || w == x) { v = w; w = t; fv = fw; fw = ft; }
            else if (ft <= fv || v == x || v == w) { v = t; fv = ft; }
        }
    }
    *ixmid_real = x;
    return -fx;
}

__global__ __launch_bounds__(64 * PI_WPB) void k_pitch_frames(
    const int16_t *__restrict__ pcm, const PiSlice *__restrict__ slices, const PiWork *__restrict__ work, int n_work,
    PiParams P, const double *__restrict__ window, const double *__restrict__ windowR,
    const long long *acc_sum, const int *acc_hi, const int *acc_lo, size_t acc_stride,
    double *__restrict__ cand /* [frames][32]: 16 freq, 16 strength */, int *__restrict__ ncand, double *__restrict__ intensity)
{
    extern __shared__ double lds[];
    // XCD-aware remap: consecutive work items (overlapping windows of one slice) go to one XCD's L2
    const int nb = (int)gridDim.x;
    int bid = (int)blockIdx.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    if (bid >= n_work) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const PiWork wk = work[bid];
    const PiSlice s = slices[wk.slice];
    const int iframe = wk.frame0 + wv;                  // 0-based
    if (iframe >= s.n_frames) return;
    double *xs = lds + (size_t)wv * (size_t)(P.xs_len + P.rr_len);
    double *rr = xs + P.xs_len;                         // rr[bix + k] = r[k], k in [-bix, bix]
    const int64_t fidx = s.frame_off + iframe;

    // global mean / peak of the slice from the exact integer accumulators
    double globalPeak;
    {
        const char *base = reinterpret_cast<const char *>(acc_sum) + acc_stride * (size_t)wk.slice;
        const long long isum = *reinterpret_cast<const long long *>(base);
        int hi = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(acc_hi) + acc_stride * (size_t)wk.slice);
        int lo = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(acc_lo) + acc_stride * (size_t)wk.slice);
        if (s.begin < 0 || s.begin + s.nx > s.clip_len) { hi = max(hi, 32769); lo = max(lo, 32768); }   // virtual zeros
        const double xmax = hi ? (double)(hi - 32769) / 32768.0 : 0.0;
        const double xmin = lo ? (double)(32768 - lo) / 32768.0 : 0.0;
        const double mean = ((double)isum / 32768.0) / (double)s.nx;
        globalPeak = fmax(fabs(xmax - mean), fabs(xmin - mean));
    }

    // frame position: Sampled_indexToX / Sampled_xToLowIndex
    const double t = s.t1 + (double)iframe * P.dt;
    const int64_t L0 = (int64_t)floor((t - s.x1) / P.dx);        // leftSample - 1 (0-based)
    const int64_t ws = L0 + 1 - P.hw;                             // first sample of the window (slice-relative)
    const int64_t m0 = L0 + 1 - P.nsp, m1 = L0 + P.nsp;           // local-mean range, inclusive

    // stage raw samples (exact in fp64) and the integer local sum
    int isum = 0;
    for (int j = lane; j < P.xs_len; j += 64) {
        int v = 0;
        if (j < P.nw) {
            const int64_t rel = ws + j, cc = s.begin + rel;
            if (rel >= 0 && rel < s.nx && cc >= 0 && cc < s.clip_len) v = (int)pcm[s.clip_off + cc];
            if (rel >= m0 && rel <= m1) isum += v;
        }
        xs[j] = (double)v / 32768.0;
    }
    isum = wave_sum_i32(isum);
    const double localMean = ((double)isum / 32768.0) / (double)(2 * P.nsp);
    double lpk = 0.0;
    const int pk0 = max(P.hw + 1 - P.hsp, 1), pk1 = min(P.hw + P.hsp, P.nw);   // 1-based inclusive
    for (int j = lane; j < P.nw; j += 64) {
        const double f = (xs[j] - localMean) * window[j];
        xs[j] = f;
        if (j + 1 >= pk0 && j + 1 <= pk1) lpk = fmax(lpk, fabs(f));
    }
    const double localPeak = wave_max_f64(lpk);
    const double inten = localPeak > globalPeak ? 1.0 : localPeak / globalPeak;

    // candidate registers: lane q holds candidate q (0 = the voiceless candidate)
    double c_f = 0.0, c_s = 0.0; int c_i = 0; int n = 1;

    if (localPeak != 0.0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // autocorrelation by direct summation; lanes own lags lane, lane+64, ... (4 per pass)
        for (int base = 0; base <= P.bix; base += 256) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            const int k0 = base + lane;
            const int jn = P.nw - base;                          // beyond this every product hits the zero padding
            const double *xk = xs + k0;
            for (int j = 0; j < jn; j++) {
                const double a = xs[j];
                a0 = fma(a, xk[j], a0);
                a1 = fma(a, xk[j + 64], a1);
                a2 = fma(a, xk[j + 128], a2);
                a3 = fma(a, xk[j + 192], a3);
            }
            if (k0 <= P.bix) rr[P.bix + k0] = a0;
            if (k0 + 64 <= P.bix) rr[P.bix + k0 + 64] = a1;
            if (k0 + 128 <= P.bix) rr[P.bix + k0 + 128] = a2;
            if (k0 + 192 <= P.bix) rr[P.bix + k0 + 192] = a3;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const double ac0 = rr[P.bix];
        for (int k = lane + 1; k <= P.bix; k += 64) {
            const double v = rr[P.bix + k] / (ac0 * windowR[k]);
            rr[P.bix + k] = v; rr[P.bix - k] = v;
        }
        if (lane == 0) rr[P.bix] = 1.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();

        const int ynx = 2 * P.bix + 1;
        const int lim = min(P.maxlag, P.bix);
        const double half_vt = 0.5 * P.voicing_thr;
        for (int base = 2; base < lim; base += 64) {
            const int i = base + lane;
            bool pred = false;
            if (i < lim) {
                const double r0 = rr[P.bix + i], rm = rr[P.bix + i - 1], rp = rr[P.bix + i + 1];
                pred = r0 > half_vt && r0 > rm && r0 >= rp;
            }
            unsigned long long mask = __ballot(pred);
            while (mask) {
                const int bpos = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int im = base + bpos;
                const double r0 = rr[P.bix + im], rm = rr[P.bix + im - 1], rp = rr[P.bix + im + 1];
                const double dr = 0.5 * (rp - rm), d2r = 2.0 * r0 - rm - rp;
                const double fmx = 1.0 / P.dx / ((double)im + dr / d2r);
                double smx = sinc_wave(rr, ynx, 1.0 / P.dx / fmx + (double)(P.bix + 1), 30, lane);
                if (smx > 1.0) smx = 1.0 / smx;
                int place = -1;
                if (n < P.maxc) {
                    place = n++;
                } else {
                    // weakest candidate so far among 1..maxc-1 (first minimum wins)
                    double ls = c_s - P.octave_cost * (log(P.min_pitch / c_f) * LOG2E_D);
                    int li = lane;
                    if (lane < 1 || lane >= P.maxc) { ls = 1e300; li = 1 << 20; }
                    for (int off = 32; off > 0; off >>= 1) {
                        const double os = __shfl_xor(ls, off, 64);
                        const int oi = __shfl_xor(li, off, 64);
                        if (os < ls || (os == ls && oi < li)) { ls = os; li = oi; }
                    }
                    double weakest = 2.0;
                    if (ls < weakest) { weakest = ls; place = li; }
                    if (smx - P.octave_cost * (log(P.min_pitch / fmx) * LOG2E_D) <= weakest) place = -1;
                }
                if (place >= 0 && lane == place) { c_f = fmx; c_s = smx; c_i = im; }
            }
        }
        // second pass: maximise the sinc interpolation around every candidate
        for (int q = 1; q < n; q++) {
            const double fq = __shfl(c_f, q, 64);
            const int iq = __shfl(c_i, q, 64);
            double xmid;
            double ymid = improve_maximum_wave(rr, ynx, iq + P.bix + 1, fq > 0.3 / P.dx ? 700 : 70, &xmid, lane);
            xmid -= (double)(P.bix + 1);
            if (ymid > 1.0) ymid = 1.0 / ymid;
            if (lane == q) { c_f = 1.0 / P.dx / xmid; c_s = ymid; }
        }
    }
    if (lane < PI_MAXC) {
        cand[fidx * 32 + lane] = lane < n ? c_f : 0.0;
        cand[fidx * 32 + 16 + lane] = lane < n ? c_s : 0.0;
    }
    if (lane == 0) { ncand[fidx] = n; intensity[fidx] = inten; }
}

// ---------------------------------------------------------------------------
// Pitch_pathFinder: one wavefront per slice
// ---------------------------------------------------------------------------
constexpr int BT_TILE = 1024;             // frames per back-tracking tile

__global__ __launch_bounds__(64) void k_pitch_path(
    const PiSlice *__restrict__ slices, PiParams P, const double *__restrict__ cand, const int *__restrict__ ncand,
    const double *__restrict__ intensity, unsigned char *__restrict__ psi /* [frames][16] */,
    double *__restrict__ f0, double *__restrict__ strength)
{
    __shared__ double t_f[PATH_TILE][PI_MAXC];      // candidate frequency
    __shared__ double t_lf[PATH_TILE][PI_MAXC];     // log2(f)
    __shared__ double t_d[PATH_TILE][PI_MAXC];      // local delta
    __shared__ int t_n[PATH_TILE];
    __shared__ double p_f[PI_MAXC], p_lf[PI_MAXC], p_d[PI_MAXC];   // previous frame
    __shared__ int p_n;
    __shared__ __attribute__((aligned(16))) unsigned char t_psi[BT_TILE][PI_MAXC];
    __shared__ unsigned char t_place[BT_TILE];
    __shared__ int s_place;
    const PiSlice s = slices[blockIdx.x];
    const int nF = s.n_frames;
    if (s.status != PCE_SLICE_OK || nF <= 0) return;
    const int lane = threadIdx.x;
    const int ic2 = lane >> 2, g = lane & 3;
    const double ceiling2 = P.ceiling;
    const double timeStepCorrection = 0.01 / P.dt;
    const double ojc = P.oj_cost * timeStepCorrection, vuc = P.vuv_cost * timeStepCorrection;
    const double *cs = cand + s.frame_off * 32;
    const int *ns = ncand + s.frame_off;
    const double *is = intensity + s.frame_off;
    unsigned char *ps = psi + s.frame_off * PI_MAXC;

    for (int f0i = 0; f0i < nF; f0i += PATH_TILE) {
        const int tn = min(PATH_TILE, nF - f0i);
        // stage the local deltas of tn frames x 16 candidates (parallel, off the recurrence)
        for (int e = lane; e < tn * PI_MAXC; e += 64) {
            const int fr = e >> 4, ic = e & 15;
            const int64_t gi = f0i + fr;
            const double f = cs[gi * 32 + ic], st = cs[gi * 32 + 16 + ic];
            const double inten = is[gi];
            double uv = P.silence_thr <= 0.0 ? 0.0 : 2.0 - inten / (P.silence_thr / (1.0 + P.voicing_thr));
            uv = P.voicing_thr + (uv > 0.0 ? uv : 0.0);
            const bool voiceless = f == 0.0 || f > ceiling2;
            t_f[fr][ic] = f;
            t_lf[fr][ic] = f > 0.0 ? log(f) * LOG2E_D : 0.0;
            t_d[fr][ic] = voiceless ? uv : st - P.octave_cost * (log(P.ceiling / f) * LOG2E_D);
            if (ic == 0) t_n[fr] = ns[gi];
        }
        __syncthreads();
        for (int fr = 0; fr < tn; fr++) {
            const int gi = f0i + fr;
            const int n2 = t_n[fr];
            const double f2 = t_f[fr][ic2], lf2 = t_lf[fr][ic2], d2 = t_d[fr][ic2];
            double best = d2; int place = 0;
            if (gi > 0) {
                best = -1e30;
                const bool cur_vl = f2 <= 0.0 || f2 >= ceiling2;
                const int pn = p_n;
                for (int k = 0; k < 4; k++) {
                    const int ic1 = g + 4 * k;
                    if (ic1 < pn) {
                        const double f1 = p_f[ic1];
                        const bool prev_vl = f1 <= 0.0 || f1 >= ceiling2;
                        double tc;
                        if (cur_vl) tc = prev_vl ? 0.0 : vuc;
                        else if (prev_vl) tc = vuc;
                        else tc = ojc * fabs(p_lf[ic1] - lf2);
                        const double value = p_d[ic1] - tc + d2;
                        if (value > best) { best = value; place = ic1; }
                    }
                }
                for (int off = 1; off <= 2; off <<= 1) {
                    const double ob = __shfl_xor(best, off, 64);
                    const int op = __shfl_xor(place, off, 64);
                    if (ob > best || (ob == best && op < place)) { best = ob; place = op; }
                }
            }
            __syncthreads();                      // every lane is done reading p_*
            if (g == 0) {
                p_d[ic2] = best; p_f[ic2] = f2; p_lf[ic2] = lf2;
                ps[(int64_t)gi * PI_MAXC + ic2] = (unsigned char)place;
                if (ic2 == 0) p_n = n2;
            }
            __syncthreads();
        }
    }
    // end of the most probable path: first maximum
    if (lane == 0) {
        int place = 0; double maximum = p_d[0];
        for (int ic = 1; ic < p_n; ic++) if (p_d[ic] > maximum) { place = ic; maximum = p_d[ic]; }
        s_place = place;
    }
    __threadfence_block();
    __syncthreads();
    // back-track through LDS tiles, last tile first
    for (int hi = nF; hi > 0;) {
        const int lo = max(0, hi - BT_TILE);
        const int cnt = hi - lo;
        for (int e = lane; e < cnt; e += 64)
            *reinterpret_cast<uint4 *>(&t_psi[e][0]) = *reinterpret_cast<const uint4 *>(ps + (int64_t)(lo + e) * PI_MAXC);
        __syncthreads();
        if (lane == 0) {
            int place = s_place;
            for (int i = cnt - 1; i >= 0; i--) { t_place[i] = (unsigned char)place; place = t_psi[i][place]; }
            s_place = place;
        }
        __syncthreads();
        for (int e = lane; e < cnt; e += 64) {
            const int64_t gi = lo + e;
            const int pl = t_place[e];
            f0[s.frame_off + gi] = cs[gi * 32 + pl];
            strength[s.frame_off + gi] = cs[gi * 32 + 16 + pl];
        }
        __syncthreads();
        hi = lo;
    }
}

// ---------------------------------------------------------------------------
// voiced median (np.median) and mean log (-> geometric mean): one workgroup per slice
// ---------------------------------------------------------------------------
struct PiSummaryDev { long long n_voiced; double median; double mean_log; };

__global__ __launch_bounds__(256) void k_pitch_median(const PiSlice *__restrict__ slices, const double *__restrict__ f0,
                                                     int npow2, PiSummaryDev *__restrict__ out)
{
    extern __shared__ double sbuf[];
    __shared__ int s_cnt;
    __shared__ double s_red[4];
    const PiSlice s = slices[blockIdx.x];
    const int tid = threadIdx.x;
    if (s.status != PCE_SLICE_OK || s.n_frames <= 0) {
        if (tid == 0) { out[blockIdx.x].n_voiced = 0; out[blockIdx.x].median = 0.0; out[blockIdx.x].mean_log = 0.0; }
        return;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const double *f = f0 + s.frame_off;
    double lsum = 0.0;
    for (int i = tid; i < s.n_frames; i += 256) {
        const double v = f[i];
        if (v > 0.0) { const int k = atomicAdd(&s_cnt, 1); sbuf[k] = v; lsum += log(v); }
    }
    __syncthreads();
    const int nv = s_cnt;
    int m = 1; while (m < nv) m <<= 1;
    if (m > npow2) m = npow2;
    for (int i = nv + tid; i < m; i += 256) sbuf[i] = __builtin_huge_val();
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < m; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const double a = sbuf[i], b = sbuf[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { sbuf[i] = b; sbuf[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) {
        PiSummaryDev o;
        o.n_voiced = nv;
        if (nv == 0) { o.median = 0.0; o.mean_log = 0.0; }
        else {
            o.median = (nv & 1) ? sbuf[nv / 2] : (sbuf[nv / 2 - 1] + sbuf[nv / 2]) / 2.0;
            o.mean_log = (((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]) / (double)nv;
        }
        out[blockIdx.x] = o;
    }
}

// Hanning window and its normalised autocorrelation (direct sums, fp64), host side.
void make_window_tables(const PitchPlan &pl, std::vector<double> &window, std::vector<double> &windowR)
{
    const int64_t nw = pl.nsamp_window, bix = pl.brent_ixmax;
    window.resize((size_t)nw);
    for (int64_t i = 1; i <= nw; i++) window[(size_t)(i - 1)] = 0.5 - 0.5 * std::cos((double)i * 2.0 * PI_D / (double)(nw + 1));
    windowR.assign((size_t)(bix + 1), 0.0);
    for (int64_t k = 0; k <= bix; k++) {
        long double acc = 0.0L;
        for (int64_t j = 0; j + k < nw; j++) acc += (long double)window[(size_t)j] * (long double)window[(size_t)(j + k)];
        windowR[(size_t)k] = (double)acc;
    }
    const double w0 = windowR[0];
    for (int64_t k = 1; k <= bix; k++) windowR[(size_t)k] /= w0;
    windowR[0] = 1.0;
}

bool same_params(const pce_pitch_params &a, const pce_pitch_params &b) { return memcmp(&a, &b, sizeof a) == 0; }

} // namespace

// ---------------------------------------------------------------------------
// host API
// ---------------------------------------------------------------------------
static int pitch_plan_slices(pce_ctx *c, const pce_pitch_params *p, const pce_slice *slices, int32_t n,
                             std::vector<int64_t> &frame_off, std::vector<int32_t> &status, std::vector<double> &t1,
                             PitchPlan *common)
{
    const double dx = 1.0 / (double)c->rate;
    frame_off.assign((size_t)n + 1, 0); status.assign((size_t)n, PCE_SLICE_OK); t1.assign((size_t)n, 0.0);
    bool have = false;
    for (int32_t i = 0; i < n; i++) {
        const pce_slice &s = slices[i];
        if (s.clip < 0 || s.clip >= c->n_clips) return pce_fail(c, PCE_E_INVALID, "slice %d: clip %d out of range", i, s.clip);
        if (s.end < s.begin) return pce_fail(c, PCE_E_INVALID, "slice %d: end < begin", i);
        PitchPlan pl;
        const int64_t nx = s.end - s.begin;
        int st = nx == 0 ? PCE_SLICE_EMPTY : pitch_plan_make(nx, dx, s.x1, p, &pl);
        status[(size_t)i] = st;
        int64_t nf = 0;
        if (st == PCE_SLICE_OK) { nf = pl.n_frames; t1[(size_t)i] = pl.t1; if (!have) { *common = pl; have = true; } }
        frame_off[(size_t)i + 1] = frame_off[(size_t)i] + nf;
    }
    if (!have) {
        // sizes that do not depend on the slice length, for an all-too-short batch
        PitchPlan pl; memset(&pl, 0, sizeof pl);
        *common = pl;
    }
    return PCE_OK;
}

extern "C" {

int pce_pitch_plan(pce_ctx *c, const pce_pitch_params *p, const pce_slice *slices, int32_t n, int64_t *frame_offsets, int32_t *status)
{
    if (!c || !p || (!slices && n > 0) || n < 0 || !frame_offsets) return PCE_E_INVALID;
    if (c->rate <= 0) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    std::vector<int64_t> fo; std::vector<int32_t> st; std::vector<double> t1; PitchPlan common;
    int rc = pitch_plan_slices(c, p, slices, n, fo, st, t1, &common);
    if (rc) return rc;
    memcpy(frame_offsets, fo.data(), sizeof(int64_t) * (size_t)(n + 1));
    if (status) memcpy(status, st.data(), sizeof(int32_t) * (size_t)n);
    return PCE_OK;
}

int pce_pitch_run(pce_ctx *c, const pce_pitch_params *p, const pce_slice *slices, int32_t n)
{
    if (!c || !p || (!slices && n > 0) || n < 0) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    PCE_HIP(c, hipSetDevice(c->device));
    if (!(c->pi_cache.same(slices, n) && c->pi_params_valid && same_params(c->pi_params, *p))) {
        c->pi_n = -1; c->pi_params_valid = false;
        PitchPlan pl;
        int rc = pitch_plan_slices(c, p, slices, n, c->pi_frame_off, c->pi_status, c->pi_t1, &pl);
        if (rc) return rc;
        const int64_t total = c->pi_frame_off[(size_t)n];
        c->pi_total_frames = total;
        PiParams P; memset(&P, 0, sizeof P);
        if (total > 0) {
            if (pl.max_candidates > PI_MAXC) return pce_fail(c, PCE_E_LIMIT, "more than %d pitch candidates per frame requested", PI_MAXC);
            P.dx = 1.0 / (double)c->rate; P.dt = pl.dt; P.min_pitch = p->pitch_floor; P.ceiling = pl.ceiling;
            P.voicing_thr = p->voicing_threshold; P.octave_cost = p->octave_cost; P.silence_thr = p->silence_threshold;
            P.oj_cost = p->octave_jump_cost; P.vuv_cost = p->voiced_unvoiced_cost;
            P.nsp = (int)pl.nsamp_period; P.hsp = (int)pl.halfnsamp_period; P.nw = (int)pl.nsamp_window; P.hw = (int)pl.halfnsamp_window;
            P.maxlag = (int)pl.maximum_lag; P.bix = (int)pl.brent_ixmax; P.maxc = (int)pl.max_candidates;
            // xs is read up to index (nw-1) + (base + 63 + 192) with base < bix+1 rounded to 256
            const int passes = (P.bix / 256) + 1;
            P.xs_len = P.nw + passes * 256 + 2; P.xs_len += P.xs_len & 1;
            P.rr_len = 2 * P.bix + 2;
            const size_t lds = sizeof(double) * (size_t)(P.xs_len + P.rr_len) * PI_WPB;
            if (lds > 160 * 1024) return pce_fail(c, PCE_E_LIMIT, "analysis window of %d samples does not fit LDS", P.nw);
            std::vector<double> window, windowR;
            make_window_tables(pl, window, windowR);
            PCE_HIP(c, c->pi_window.reserve(sizeof(double) * window.size()));
            PCE_HIP(c, c->pi_windowR.reserve(sizeof(double) * windowR.size()));
            PCE_HIP(c, hipMemcpyAsync(c->pi_window.p, window.data(), sizeof(double) * window.size(), hipMemcpyHostToDevice, c->stream));
            PCE_HIP(c, hipMemcpyAsync(c->pi_windowR.p, windowR.data(), sizeof(double) * windowR.size(), hipMemcpyHostToDevice, c->stream));
            PCE_HIP(c, hipStreamSynchronize(c->stream));
        }
        static_assert(sizeof(PiParams) <= sizeof(c->pi_P), "PiParams storage");
        memcpy(c->pi_P, &P, sizeof P);
        // slice table + work list (one block per PI_WPB frames)
        std::vector<PiSlice> hs((size_t)(n > 0 ? n : 1));
        std::vector<PiWork> work;
        int64_t max_frames = 0;
        for (int32_t i = 0; i < n; i++) {
            const pce_slice &s = slices[i];
            PiSlice &h = hs[(size_t)i];
            h.begin = s.begin; h.clip_off = c->clip_off[s.clip]; h.clip_len = c->clip_off[s.clip + 1] - c->clip_off[s.clip];
            h.nx = s.end - s.begin; h.frame_off = c->pi_frame_off[(size_t)i]; h.x1 = s.x1; h.t1 = c->pi_t1[(size_t)i];
            const int64_t nf = c->pi_frame_off[(size_t)i + 1] - c->pi_frame_off[(size_t)i];
            if (nf > INT32_MAX) return pce_fail(c, PCE_E_LIMIT, "slice %d has too many frames", i);
            h.n_frames = (int32_t)nf; h.status = c->pi_status[(size_t)i];
            if (nf > max_frames) max_frames = nf;
            for (int64_t f = 0; f < nf; f += PI_WPB) work.push_back({i, (int32_t)f});
        }
        int np2 = 1; while (np2 < max_frames) np2 <<= 1;
        if ((size_t)np2 * sizeof(double) > 128 * 1024) return pce_fail(c, PCE_E_LIMIT, "slice with %lld frames exceeds the %d-frame median limit", (long long)max_frames, 16384);
        c->pi_np2 = np2;
        c->pi_n_work = (int64_t)work.size();
        PCE_HIP(c, c->pi_meta.reserve(sizeof(PiSlice) * hs.size()));
        PCE_HIP(c, c->pi_work.reserve(sizeof(PiWork) * (work.size() + 1)));
        PCE_HIP(c, c->pi_cand.reserve(sizeof(double) * 32 * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_gpeak.reserve((sizeof(int) + sizeof(double)) * (size_t)(total + 1)));   // ncand + intensity
        PCE_HIP(c, c->pi_psi.reserve((size_t)PI_MAXC * (size_t)(total + 1) + 16));
        PCE_HIP(c, c->pi_f0.reserve(sizeof(double) * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_strength.reserve(sizeof(double) * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_summary.reserve(sizeof(PiSummaryDev) * hs.size()));
        PCE_HIP(c, hipMemcpyAsync(c->pi_meta.p, hs.data(), sizeof(PiSlice) * hs.size(), hipMemcpyHostToDevice, c->stream));
        if (!work.empty())
            PCE_HIP(c, hipMemcpyAsync(c->pi_work.p, work.data(), sizeof(PiWork) * work.size(), hipMemcpyHostToDevice, c->stream));
        PCE_HIP(c, hipStreamSynchronize(c->stream));
        rc = pce_energy_plan(c, slices, n, c->pi_peakwork, c->pi_acc, &c->pi_n_energy_work);
        if (rc) return rc;
        c->pi_cache.store(slices, n);
        c->pi_params = *p; c->pi_params_valid = true;
    }
    PiParams P; memcpy(&P, c->pi_P, sizeof P);
    const int64_t total = c->pi_total_frames;
    if (total > 0) {
        int rc = pce_energy_launch(c, n, 500, c->pi_n_energy_work, c->pi_peakwork, c->pi_acc);
        if (rc) return rc;
        size_t stride; const long long *a_sum; const int *a_hi, *a_lo;
        pce_energy_range_ptrs(c->pi_acc, &stride, &a_sum, &a_hi, &a_lo);
        double *intensity = c->pi_gpeak.as<double>();
        int *ncand = reinterpret_cast<int *>(intensity + (total + 1));
        {
            int64_t nb = c->pi_n_work;
            nb = (nb + 7) & ~(int64_t)7;         // multiple of 8 so the XCD remap is a bijection; extra blocks exit
            const size_t lds = sizeof(double) * (size_t)(P.xs_len + P.rr_len) * PI_WPB;
            if (lds > 64 * 1024)
                PCE_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pitch_frames), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            KernelTimer t(c, PCE_K_PITCH_FRAMES);
            hipLaunchKernelGGL(k_pitch_frames, dim3((unsigned)nb), dim3(64 * PI_WPB), lds, c->stream, c->d_pcm,
                               c->pi_meta.as<PiSlice>(), c->pi_work.as<PiWork>(), (int)c->pi_n_work, P,
                               c->pi_window.as<double>(), c->pi_windowR.as<double>(), a_sum, a_hi, a_lo, stride,
                               c->pi_cand.as<double>(), ncand, intensity);
        }
        {
            KernelTimer t(c, PCE_K_PITCH_PATH);
            hipLaunchKernelGGL(k_pitch_path, dim3((unsigned)n), dim3(64), 0, c->stream, c->pi_meta.as<PiSlice>(), P,
                               c->pi_cand.as<double>(), ncand, intensity, c->pi_psi.as<unsigned char>(),
                               c->pi_f0.as<double>(), c->pi_strength.as<double>());
        }
    }
    if (n > 0) {
        const size_t lds = sizeof(double) * (size_t)c->pi_np2;
        if (lds > 64 * 1024)
            PCE_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pitch_median), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KernelTimer t(c, PCE_K_PITCH_MEDIAN);
        hipLaunchKernelGGL(k_pitch_median, dim3((unsigned)n), dim3(256), lds, c->stream, c->pi_meta.as<PiSlice>(),
                           c->pi_f0.as<double>(), c->pi_np2, c->pi_summary.as<PiSummaryDev>());
    }
    PCE_HIP(c, hipGetLastError());
    c->pi_n = n;
    return PCE_OK;
}

int pce_pitch_fetch(pce_ctx *c, double *f0, double *strength, pce_pitch_summary *summary)
{
    if (!c) return PCE_E_INVALID;
    if (c->pi_n < 0) return pce_fail(c, PCE_E_STATE, "pce_pitch_fetch before pce_pitch_run");
    PCE_HIP(c, hipSetDevice(c->device));
    const int32_t n = c->pi_n;
    const int64_t total = c->pi_total_frames;
    std::vector<PiSummaryDev> sd((size_t)(n > 0 ? n : 1));
    if (f0 && total > 0) PCE_HIP(c, hipMemcpyAsync(f0, c->pi_f0.p, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
    if (strength && total > 0) PCE_HIP(c, hipMemcpyAsync(strength, c->pi_strength.p, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, c->stream));
    if (summary && n > 0) PCE_HIP(c, hipMemcpyAsync(sd.data(), c->pi_summary.p, sizeof(PiSummaryDev) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    if (summary)
        for (int32_t i = 0; i < n; i++) {
            pce_pitch_summary &o = summary[i];
            o.n_frames = c->pi_frame_off[(size_t)i + 1] - c->pi_frame_off[(size_t)i];
            o.n_voiced = sd[(size_t)i].n_voiced; o.median_f0 = sd[(size_t)i].median; o.mean_log_f0 = sd[(size_t)i].mean_log;
            o.t1 = c->pi_t1[(size_t)i]; o.status = c->pi_status[(size_t)i]; o.reserved = 0;
        }
    return PCE_OK;
}

} // extern "C"
