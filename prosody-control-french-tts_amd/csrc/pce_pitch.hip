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
#include <cstdlib>

int pce_energy_plan(pce_ctx *c, const pce_slice *slices, int32_t n, DevBuf &work_buf, DevBuf &out_buf, int64_t *n_work);
int pce_energy_launch(pce_ctx *c, int32_t n, int32_t loud_thr, int64_t n_work, DevBuf &work_buf, DevBuf &out_buf, hipStream_t on = nullptr);
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
constexpr int PI_WPB = 4;                 // waves per block in k_pitch_frames
constexpr int PI_FPB = 4;                 // frames per work item (one per wave measured fastest: silent frames exit early)
constexpr int PI_MAXC = 16;               // candidates per frame the kernels can hold
constexpr int PI_MEDIAN_LDS = 16384;      // voiced F0 values k_pitch_median sorts in LDS (128 KB)
constexpr int RF_LISTS = 256;             // independent candidate lists (one hot counter would serialise in L2)
constexpr int RF_CSTRIDE = 32;            // counters on their own 128-byte lines             // frames staged per LDS tile in k_pitch_path

struct PiParams {
    double dx, dt, min_pitch, ceiling, voicing_thr, octave_cost, silence_thr, oj_cost, vuv_cost;
    int nsp, hsp, nw, hw, maxlag, bix, maxc, nfft, zlen, rr_len, mode, fpb;
    int o_tw2, o_twN, o_win, o_winR, blob_f64, pcm_span, tabs, rr_half;   // register paths: table blob layout (offsets in doubles)
    int refine_seeded, pad_;                                              // k_pitch_refine: 1 = parabolic search first (default), 0 = Praat's own iterates for every candidate
    double refine_tol_rel;                                                // ... and the relative step below which the parabolic search stops (3e-8)
};
struct PiSlice {
    int64_t begin, clip_len, clip_off, nx, frame_off;
    double x1, t1;
    int32_t n_frames, status;
};
struct PiWork { int32_t slice, frame0; };
struct RefineItem { long long frame; int slot; int imax; };

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

// ---------------------------------------------------------------------------
// fp64 autocorrelation by FFT, one wavefront per frame, entirely in LDS.
// The zero-padded real frame x[0..N) is read as M = N/2 complex points z[n] = x[2n] + i x[2n+1]
// (the staging buffer IS that array), transformed with Stockham radix-4 (+ one radix-2)
// passes that ping-pong between two LDS buffers, untangled into the power spectrum
// P[k] = |X[k]|^2, re-tangled (P is real and even) and sent through the same forward
// transform again: ac[2n] = Re Y[n], ac[2n+1] = -Im Y[n] (common positive scale dropped;
// it cancels in r[k] = ac[k] / (ac[0] windowR[k])).  Praat itself takes this route
// (NUMfft_forward / power / NUMfft_backward in Sound_to_Pitch.cpp); N = nsampFFT.
// LDS index padding: one complex per 8 keeps the stride-4 Stockham stores conflict-free.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double rcp_f64(double a)
{
    // v_rcp_f64 is good to 2^-24 (measured 4.6e-8); one cubic step r (1 + e + e^2) leaves e^3 ~ 1e-22
    const double r = __builtin_amdgcn_rcp(a);
    const double e = fma(-a, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
__device__ __forceinline__ int ZP(int p) { return p + (p >> 3); }
__device__ __forceinline__ double2 cmul_f64(double2 a, double2 w)
{
    return make_double2(fma(a.x, w.x, -(a.y * w.y)), fma(a.x, w.y, a.y * w.x));
}
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// forward DFT of M points (exp(-2 pi i ...)); returns the buffer holding the result
__device__ double2 *fft_wave(double2 *src, double2 *dst, int M, const double2 *__restrict__ twM, int lane)
{
    int Ns = 1;
    while (Ns < M) {
        if (M / Ns >= 4) {
            const int nb = M >> 2, tws = M / (Ns << 2);
            for (int b = lane; b < nb; b += 64) {
                const int k = b & (Ns - 1);
                double2 a0 = src[ZP(b)], a1 = src[ZP(b + nb)], a2 = src[ZP(b + 2 * nb)], a3 = src[ZP(b + 3 * nb)];
                if (Ns > 1) {
                    a1 = cmul_f64(a1, twM[k * tws]);
                    a2 = cmul_f64(a2, twM[2 * k * tws]);
                    a3 = cmul_f64(a3, twM[3 * k * tws]);
                }
                const double2 b0 = make_double2(a0.x + a2.x, a0.y + a2.y), b1 = make_double2(a0.x - a2.x, a0.y - a2.y);
                const double2 b2 = make_double2(a1.x + a3.x, a1.y + a3.y);
                const double2 b3 = make_double2(a1.y - a3.y, -(a1.x - a3.x));          // (a1 - a3) * (-i)
                const int base = ((b - k) << 2) + k;
                dst[ZP(base)] = make_double2(b0.x + b2.x, b0.y + b2.y);
                dst[ZP(base + Ns)] = make_double2(b1.x + b3.x, b1.y + b3.y);
                dst[ZP(base + 2 * Ns)] = make_double2(b0.x - b2.x, b0.y - b2.y);
                dst[ZP(base + 3 * Ns)] = make_double2(b1.x - b3.x, b1.y - b3.y);
            }
            Ns <<= 2;
        } else {
            const int nb = M >> 1, tws = M / (Ns << 1);
            for (int b = lane; b < nb; b += 64) {
                const int k = b & (Ns - 1);
                const double2 a0 = src[ZP(b)];
                double2 a1 = src[ZP(b + nb)];
                if (Ns > 1) a1 = cmul_f64(a1, twM[k * tws]);
                const int base = ((b - k) << 1) + k;
                dst[ZP(base)] = make_double2(a0.x + a1.x, a0.y + a1.y);
                dst[ZP(base + Ns)] = make_double2(a0.x - a1.x, a0.y - a1.y);
            }
            Ns <<= 1;
        }
        wave_sync();
        double2 *t = src; src = dst; dst = t;
    }
    return src;
}


// ---------------------------------------------------------------------------
// M = 512 (16 kHz at Praat's 75 Hz floor: N = 1024) keeps the whole transform in registers:
// 512 = 8 x 8 x 8, eight complex points per lane, three radix-8 butterflies per lane and two
// transposes through ONE 9 KB LDS exchange buffer per wavefront.  LDS stores are the scarce
// resource here (ds_write_b128: ~79 B/clk/CU against 256 B/clk for reads), and this form
// stores 40 KB per frame where the ping-pong Stockham passes store ~115 KB; the single buffer
// also doubles the wavefronts a CU can hold.
//   stage 1: lane l holds x[64 a + l]        -> DFT over a, twiddle W512^(l k0)
//   stage 2: lane (c + 8 k0) holds y1[k0][8 b + c] -> DFT over b, twiddle W64^(c k1)
//   stage 3: lane (k0 + 8 k1) holds y2[k0][k1][c]  -> DFT over c: X[k0 + 8 k1 + 64 k2]
// so the result comes back in the layout stage 1 started from (index = lane + 64 r).
// Exchange images: [k0][72] and [c][65] complex: ds_write_b128 (8 contiguous lanes per group,
// 128-byte bank rows) and ds_read_b128 (16-lane groups, 256-byte rows) are both conflict-free.
// ---------------------------------------------------------------------------
constexpr int R_WAVE_F64 = 2 * 8 * 72;              // doubles of LDS per wavefront on the register paths (MODE 1, 2)
constexpr int R3_WAVE_F64 = 2 * 1168;               // ... MODE 3: images [16][72] and [8][129] (+64) complex
constexpr int reg_wave_f64(int mode) { return mode == 3 ? R3_WAVE_F64 : R_WAVE_F64; }
__device__ __forceinline__ void dft8_f64(double2 (&a)[8])
{
    const double r = 0.70710678118654752440;
    const double2 b0 = make_double2(a[0].x + a[4].x, a[0].y + a[4].y), b4 = make_double2(a[0].x - a[4].x, a[0].y - a[4].y);
    const double2 b1 = make_double2(a[1].x + a[5].x, a[1].y + a[5].y), t5 = make_double2(a[1].x - a[5].x, a[1].y - a[5].y);
    const double2 b2 = make_double2(a[2].x + a[6].x, a[2].y + a[6].y), t6 = make_double2(a[2].x - a[6].x, a[2].y - a[6].y);
    const double2 b3 = make_double2(a[3].x + a[7].x, a[3].y + a[7].y), t7 = make_double2(a[3].x - a[7].x, a[3].y - a[7].y);
    const double2 b5 = make_double2(r * (t5.x + t5.y), r * (t5.y - t5.x));        // * exp(-i pi/4)
    const double2 b6 = make_double2(t6.y, -t6.x);                                 // * (-i)
    const double2 b7 = make_double2(r * (t7.y - t7.x), -(r * (t7.x + t7.y)));     // * exp(-3 i pi/4)
    const double2 c0 = make_double2(b0.x + b2.x, b0.y + b2.y), c2 = make_double2(b0.x - b2.x, b0.y - b2.y);
    const double2 c1 = make_double2(b1.x + b3.x, b1.y + b3.y), c3 = make_double2(b1.y - b3.y, -(b1.x - b3.x));
    const double2 d0 = make_double2(b4.x + b6.x, b4.y + b6.y), d2 = make_double2(b4.x - b6.x, b4.y - b6.y);
    const double2 d1 = make_double2(b5.x + b7.x, b5.y + b7.y), d3 = make_double2(b5.y - b7.y, -(b5.x - b7.x));
    a[0] = make_double2(c0.x + c1.x, c0.y + c1.y); a[4] = make_double2(c0.x - c1.x, c0.y - c1.y);
    a[2] = make_double2(c2.x + c3.x, c2.y + c3.y); a[6] = make_double2(c2.x - c3.x, c2.y - c3.y);
    a[1] = make_double2(d0.x + d1.x, d0.y + d1.y); a[5] = make_double2(d0.x - d1.x, d0.y - d1.y);
    a[3] = make_double2(d2.x + d3.x, d2.y + d3.y); a[7] = make_double2(d2.x - d3.x, d2.y - d3.y);
}
// forward DFT of 512 points held as z[r] = x[lane + 64 r]; returns X[lane + 64 r] in z[r]
__device__ __forceinline__ void fft512_reg(double2 (&z)[8], double2 *ex, const double2 *tw1 /* [7][64]: W512^(l k) */, const double2 *tw2 /* [7][8]: W64^(c k) */, int lane)
{
    dft8_f64(z);
#pragma unroll
    for (int k = 1; k < 8; k++) z[k] = cmul_f64(z[k], tw1[(k - 1) * 64 + lane]);
#pragma unroll
    for (int k = 0; k < 8; k++) ex[72 * k + lane] = z[k];
    wave_sync();
    const int c = lane & 7, k0 = lane >> 3;
#pragma unroll
    for (int b = 0; b < 8; b++) z[b] = ex[72 * k0 + 8 * b + c];
    wave_sync();
    dft8_f64(z);
#pragma unroll
    for (int k = 1; k < 8; k++) z[k] = cmul_f64(z[k], tw2[(k - 1) * 8 + c]);
#pragma unroll
    for (int k = 0; k < 8; k++) ex[65 * c + k0 + 8 * k] = z[k];
    wave_sync();
#pragma unroll
    for (int q = 0; q < 8; q++) z[q] = ex[65 * q + lane];
    wave_sync();
    dft8_f64(z);
}

// forward DFT of 1024 points (N = 2048: 44.1 kHz at the 150 Hz floor -- the rate of the reference's own recordings)
// by one wavefront, 16 complex points per lane: 1024 = 16 x 8 x 8.
//   stage 1: lane l holds x[64 a + l], a < 16         -> DFT16 over a, twiddle W1024^(l k0)
//   stage 2: lane (c + 8 k0lo) holds y1[k0lo + 8 g][8 b + c]  -> two DFT8 over b, twiddle W64^(c k1)
//   stage 3: lane (k0 + 16 k1lo) holds y2[k0][k1lo + 4 g][c]  -> two DFT8 over c: X[k0 + 16 k1 + 128 k2]
// and the result is back in the input layout (index = lane + 64 r, r = g + 2 k2).  Exchange images [k0][72] and
// [c][129] complex (conflict-free for ds_write_b128 / ds_read_b128 like the 512-point ones).
__device__ __forceinline__ void dft4_f64(double2 &a0, double2 &a1, double2 &a2, double2 &a3)
{
    const double2 t0 = make_double2(a0.x + a2.x, a0.y + a2.y), t1 = make_double2(a0.x - a2.x, a0.y - a2.y);
    const double2 t2 = make_double2(a1.x + a3.x, a1.y + a3.y), t3 = make_double2(a1.y - a3.y, -(a1.x - a3.x));   // (a1 - a3)(-i)
    a0 = make_double2(t0.x + t2.x, t0.y + t2.y); a1 = make_double2(t1.x + t3.x, t1.y + t3.y);
    a2 = make_double2(t0.x - t2.x, t0.y - t2.y); a3 = make_double2(t1.x - t3.x, t1.y - t3.y);
}
__device__ __forceinline__ void dft16_f64(double2 (&x)[16])
{
    // n = 4 n1 + n2: DFT4 over n1, twiddle W16^(n2 k1), DFT4 over n2 -> X[k1 + 4 k2]
    const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, r = 0.70710678118654752440;
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4_f64(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);       // x[4 k1 + n2] = A[n2][k1]
    auto mulc = [](double2 a, double wr, double wi) { return make_double2(fma(a.x, wr, -(a.y * wi)), fma(a.x, wi, a.y * wr)); };
    x[4 + 1] = mulc(x[4 + 1], c1, -s1);  x[4 + 2] = mulc(x[4 + 2], r, -r);   x[4 + 3] = mulc(x[4 + 3], s1, -c1);     // W16^1, W16^2, W16^3
    x[8 + 1] = mulc(x[8 + 1], r, -r);    x[8 + 2] = make_double2(x[8 + 2].y, -x[8 + 2].x);                           // W16^2, W16^4 = -i
    x[8 + 3] = mulc(x[8 + 3], -r, -r);                                                                             // W16^6
    x[12 + 1] = mulc(x[12 + 1], s1, -c1); x[12 + 2] = mulc(x[12 + 2], -r, -r); x[12 + 3] = mulc(x[12 + 3], -c1, s1); // W16^3, W16^6, W16^9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4_f64(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);   // x[4 k1 + k2] = X[k1 + 4 k2]
    // to natural order X[k] at x[k]: transpose the 4 x 4 index (k1, k2) -> (k2, k1)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = i + 1; j < 4; j++) { const double2 t = x[4 * i + j]; x[4 * i + j] = x[4 * j + i]; x[4 * j + i] = t; }
}
__device__ __forceinline__ void fft1024_reg(double2 (&z)[16], double2 *ex, const double2 *tw1 /* [15][64]: W1024^(l k) */,
                                            const double2 *tw2 /* [7][8]: W64^(c k) */, int lane)
{
    dft16_f64(z);
#pragma unroll
    for (int k = 1; k < 16; k++) z[k] = cmul_f64(z[k], tw1[(k - 1) * 64 + lane]);
#pragma unroll
    for (int k = 0; k < 16; k++) ex[72 * k + lane] = z[k];
    wave_sync();
    const int c = lane & 7, k0lo = lane >> 3;
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int b = 0; b < 8; b++) z[8 * g + b] = ex[72 * (k0lo + 8 * g) + 8 * b + c];
    wave_sync();
    {
        double2 h0[8], h1[8];
#pragma unroll
        for (int b = 0; b < 8; b++) { h0[b] = z[b]; h1[b] = z[8 + b]; }
        dft8_f64(h0); dft8_f64(h1);
#pragma unroll
        for (int k = 1; k < 8; k++) { const double2 w = tw2[(k - 1) * 8 + c]; h0[k] = cmul_f64(h0[k], w); h1[k] = cmul_f64(h1[k], w); }
#pragma unroll
        for (int k = 0; k < 8; k++) { ex[129 * c + k0lo + 16 * k] = h0[k]; ex[129 * c + k0lo + 8 + 16 * k] = h1[k]; }
    }
    wave_sync();
    {
        double2 h0[8], h1[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { h0[q] = ex[129 * q + lane]; h1[q] = ex[129 * q + 64 + lane]; }
        wave_sync();
        dft8_f64(h0); dft8_f64(h1);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) { z[2 * k2] = h0[k2]; z[2 * k2 + 1] = h1[k2]; }
    }
}

// forward DFT of 256 points by a HALF wavefront (two frames per wavefront): 256 = 8 x 8 x 4,
// z[r] = x[hl + 32 r] in, X[hl + 32 r] out, hl = lane within the half.  Exchange images
// [k0][36] and [c][66] complex in this frame's 288-complex region (conflict-free as above).
__device__ __forceinline__ void fft256_half(double2 (&z)[8], double2 *ex, const double2 *tw1 /* [7][32]: W256^(l k) */, const double2 *tw2 /* [7][4]: W32^(c k) */, int hl)
{
    dft8_f64(z);
#pragma unroll
    for (int k = 1; k < 8; k++) z[k] = cmul_f64(z[k], tw1[(k - 1) * 32 + hl]);
#pragma unroll
    for (int k = 0; k < 8; k++) ex[36 * k + hl] = z[k];
    wave_sync();
    const int c = hl & 3, k0 = hl >> 2;
#pragma unroll
    for (int b = 0; b < 8; b++) z[b] = ex[36 * k0 + 4 * b + c];
    wave_sync();
    dft8_f64(z);
#pragma unroll
    for (int k = 1; k < 8; k++) z[k] = cmul_f64(z[k], tw2[(k - 1) * 4 + c]);
#pragma unroll
    for (int k = 0; k < 8; k++) ex[66 * c + k0 + 8 * k] = z[k];
    wave_sync();
    double2 v[2][4];
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
        for (int q = 0; q < 4; q++) v[g][q] = ex[66 * q + 32 * g + hl];
    wave_sync();
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const double2 t0 = make_double2(v[g][0].x + v[g][2].x, v[g][0].y + v[g][2].y), t1 = make_double2(v[g][0].x - v[g][2].x, v[g][0].y - v[g][2].y);
        const double2 t2 = make_double2(v[g][1].x + v[g][3].x, v[g][1].y + v[g][3].y), t3 = make_double2(v[g][1].y - v[g][3].y, -(v[g][1].x - v[g][3].x));
        z[g] = make_double2(t0.x + t2.x, t0.y + t2.y);     z[2 + g] = make_double2(t1.x + t3.x, t1.y + t3.y);
        z[4 + g] = make_double2(t0.x - t2.x, t0.y - t2.y); z[6 + g] = make_double2(t1.x - t3.x, t1.y - t3.y);
    }
}

template <int W> __device__ __forceinline__ int group_sum_i32(int v)
{
    for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int W> __device__ __forceinline__ double group_max_f64(double v)
{
    for (int off = W / 2; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// MODE 0: any N, one frame per wavefront, Stockham passes between two LDS buffers.
// MODE 1: N = 1024, one frame per wavefront, register-resident transform.
// MODE 2: N = 512 (16 kHz at the reference's 150 Hz floor, Code/audioPipeline.py:329), TWO frames per
//         wavefront (one per 32-lane half), register-resident transform.
// MODE 3: N = 2048 (44.1 kHz at that floor), one frame per wavefront, 16 points per lane.
template <int WPB, int MODE, bool TABS>
__global__ __launch_bounds__(64 * WPB) void k_pitch_frames(
    const int16_t *__restrict__ pcm, const PiSlice *__restrict__ slices, const PiWork *__restrict__ work, int n_work,
    PiParams P, const double *__restrict__ window, const double *__restrict__ windowR,
    const double2 *__restrict__ twM /* exp(-2 pi i m / M), m < M */, const double2 *__restrict__ twN /* exp(-2 pi i k / N), k <= M */,
    const long long *acc_sum, const int *acc_hi, const int *acc_lo, size_t acc_stride,
    double *__restrict__ cand /* [frames][32]: 16 freq, 16 strength */, int *__restrict__ ncand, double *__restrict__ intensity,
    double *__restrict__ rr_out /* [frames][rr_half]: r[0..bix] */, RefineItem *__restrict__ items, unsigned int *__restrict__ item_count,
    unsigned int list_cap, const double *__restrict__ blob /* register paths: lane-ordered twiddles, twN, window, windowR */)
{
    // the LUFS chain runs beside this kernel on a side stream (few waves, long dependent fp64 chains): with this kernel's
    // waves at a higher issue priority it only takes the slots they leave (2.94 -> 2.88 ms per step; raising the
    // priority of the path / median kernels beside the STFT pass changed nothing)
    __builtin_amdgcn_s_setprio(2);
    constexpr int LW = MODE == 2 ? 32 : 64;              // lanes per frame
    constexpr int FPW = 64 / LW;                         // frames per wavefront
    constexpr int FPB = MODE == 2 ? 2 * PI_FPB : PI_FPB; // frames per work item (host: P.fpb)
    constexpr int MR = (MODE == 3 ? 16 : 8) * LW;        // M on the register paths
    constexpr int R = MODE == 3 ? 16 : 8;                // complex points per lane on the register paths
    constexpr int RW = reg_wave_f64(MODE);               // doubles of LDS per wavefront there
    constexpr int REG_C = RW / 2 / FPW;                  // complex slots per frame on the register paths
    extern __shared__ double lds[];
    // XCD-aware remap: consecutive work items (overlapping windows of one slice) go to one XCD's L2
    const int nb = (int)gridDim.x;
    int bid = (int)blockIdx.x;
    if ((nb & 7) == 0) bid = (bid & 7) * (nb >> 3) + (bid >> 3);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hl = lane & (LW - 1), half = lane / LW;    // lane within the frame's group; the group
    const int M = P.nfft >> 1;
    if (bid >= n_work) return;
    const PiWork wk = work[bid];
    const PiSlice s = slices[wk.slice];
    double2 *wave_base = reinterpret_cast<double2 *>(lds) + (MODE == 0 ? (size_t)wv * (size_t)(2 * P.zlen) : (size_t)wv * (reg_wave_f64(MODE) / 2));
    double2 *bufA = MODE == 0 ? wave_base : wave_base + half * REG_C;
    double2 *bufB = bufA + P.zlen;                      // (MODE 0 only)
    double *xr = reinterpret_cast<double *>(bufA);      // MODE 0: real view of bufA: x[j] at xr[2 ZP(j>>1) + (j&1)]
    // register paths: the tables (optionally) and the samples under this work item's frames go to LDS once
    // per workgroup -- per-lane 2-byte and gather loads were the bottleneck (vector-memory issue, not bytes)
    const double2 *tw1 = nullptr, *tw2 = nullptr, *twR = nullptr;
    const double *win = nullptr, *winR = nullptr;
    const int16_t *spcm = nullptr;
    int64_t lo = 0;
    if constexpr (MODE != 0) {
        double *tab = lds + WPB * RW;
        if constexpr (TABS) {
            for (int i = (int)threadIdx.x; i < (P.blob_f64 >> 1); i += 64 * WPB)
                reinterpret_cast<double2 *>(tab)[i] = reinterpret_cast<const double2 *>(blob)[i];
            tw1 = reinterpret_cast<const double2 *>(tab); tw2 = reinterpret_cast<const double2 *>(tab + P.o_tw2);
            twR = reinterpret_cast<const double2 *>(tab + P.o_twN); win = tab + P.o_win; winR = tab + P.o_winR;
        } else {
            tw1 = reinterpret_cast<const double2 *>(blob); tw2 = reinterpret_cast<const double2 *>(blob + P.o_tw2);
            twR = reinterpret_cast<const double2 *>(blob + P.o_twN); win = blob + P.o_win; winR = blob + P.o_winR;
        }
    }
    {   // every mode: the samples under this work item's frames, staged once per workgroup
        int16_t *sp = MODE != 0 ? reinterpret_cast<int16_t *>(lds + WPB * RW + (TABS ? P.blob_f64 : 0))
                                : reinterpret_cast<int16_t *>(lds + (size_t)WPB * 4 * (size_t)P.zlen);
        const double tA = s.t1 + (double)wk.frame0 * P.dt;
        lo = (int64_t)floor((tA - s.x1) / P.dx) + 1 - P.hw;       // window start of the first frame (slice-relative)
        for (int i = (int)threadIdx.x; i < P.pcm_span; i += 64 * WPB) {
            const int64_t rel = lo + i, cc = s.begin + rel;
            int v = 0;
            if (rel >= 0 && rel < s.nx && cc >= 0 && cc < s.clip_len) v = (int)pcm[s.clip_off + cc];
            sp[i] = (int16_t)v;
        }
        spcm = sp;
        __syncthreads();
    }
    for (int fi = wv * FPW; fi < FPB; fi += WPB * FPW) {
    const int iframe = wk.frame0 + fi + half;           // 0-based
    const bool live = iframe < s.n_frames;
    if (__ballot(live) == 0) break;
    const int64_t fidx = s.frame_off + iframe;
    wave_sync();                                        // the previous frame's LDS reads are done

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

    double2 z[R];                                        // register paths: z[r] = x[2n] + i x[2n+1], n = hl + LW r
    double lpk = 0.0;
    const int pk0 = max(P.hw + 1 - P.hsp, 1), pk1 = min(P.hw + P.hsp, P.nw);   // 1-based inclusive
    if constexpr (MODE != 0) {
        int isum = 0, v[2 * R];
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = 2 * (hl + LW * r) + h;
                int x = 0;
                if (live && j < P.nw) {
                    const int64_t rel = ws + j;
                    x = (int)spcm[(int)(ws - lo) + j];
                    if (rel >= m0 && rel <= m1) isum += x;
                }
                v[2 * r + h] = x;
            }
        }
        isum = group_sum_i32<LW>(isum);
        const double localMean = ((double)isum / 32768.0) / (double)(2 * P.nsp);
#pragma unroll
        for (int r = 0; r < R; r++) {
            double f[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = 2 * (hl + LW * r) + h;
                f[h] = 0.0;
                if (live && j < P.nw) {
                    f[h] = ((double)v[2 * r + h] / 32768.0 - localMean) * win[j];
                    if (j + 1 >= pk0 && j + 1 <= pk1) lpk = fmax(lpk, fabs(f[h]));
                }
            }
            z[r] = make_double2(f[0], f[1]);
        }
    } else {
    // stage raw samples (exact in fp64) and the integer local sum
    int isum = 0;
    for (int j = lane; j < P.nfft; j += 64) {
        int v = 0;
        if (j < P.nw) {
            const int64_t rel = ws + j;
            v = (int)spcm[(int)(ws - lo) + j];
            if (rel >= m0 && rel <= m1) isum += v;
        }
        xr[2 * ZP(j >> 1) + (j & 1)] = (double)v / 32768.0;
    }
    isum = wave_sum_i32(isum);
    const double localMean = ((double)isum / 32768.0) / (double)(2 * P.nsp);
    for (int j = lane; j < P.nw; j += 64) {
        const int a = 2 * ZP(j >> 1) + (j & 1);
        const double f = (xr[a] - localMean) * window[j];
        xr[a] = f;
        if (j + 1 >= pk0 && j + 1 <= pk1) lpk = fmax(lpk, fabs(f));
    }
    }
    const double localPeak = group_max_f64<LW>(lpk);
    const double inten = localPeak > globalPeak ? 1.0 : localPeak / globalPeak;
    const bool active = live && localPeak != 0.0;

    double *rr = nullptr;                                // rr[bix + k] = r[k], k in [-bix, bix]
    // candidate registers: group lane q holds candidate q (0 = the voiceless candidate)
    double c_f = 0.0, c_s = 0.0; int c_i = 0; int n = 1;

    if (__ballot(active) != 0) {
        if constexpr (MODE != 0) {
            wave_sync();
            if constexpr (MODE == 1) fft512_reg(z, bufA, tw1, tw2, lane); else if constexpr (MODE == 3) fft1024_reg(z, bufA, tw1, tw2, lane); else fft256_half(z, bufA, tw1, tw2, hl);
            // power spectrum of the real frame, re-tangled for the second transform.  Lane-local form of
            // the pairwise loop of MODE 0: for every own k, with zm = Z[M - k],
            // X[k] = ez + t, X[M - k]* = ez - t, and W[k] = (e - d sin, -d cos) holds for all k in [0, M).
#pragma unroll
            for (int r = 0; r < R; r++) bufA[hl + LW * r] = z[r];
            wave_sync();
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int k = hl + LW * r;
                const double2 zk = z[r], zm = bufA[(MR - k) & (MR - 1)];
                const double2 w = twR[k];                                     // (cos, -sin)
                const double2 ez = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
                const double2 oz = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
                const double2 t = cmul_f64(oz, w);
                const double xr1 = ez.x + t.x, xi1 = ez.y + t.y, xr2 = ez.x - t.x, xi2 = ez.y - t.y;
                const double pk = fma(xr1, xr1, xi1 * xi1), pm = fma(xr2, xr2, xi2 * xi2);
                const double e = 0.5 * (pk + pm), d = 0.5 * (pk - pm);
                z[r] = make_double2(e - d * (-w.y), -(d * w.x));
            }
            wave_sync();
            if constexpr (MODE == 1) fft512_reg(z, bufA, tw1, tw2, lane); else if constexpr (MODE == 3) fft1024_reg(z, bufA, tw1, tw2, lane); else fft256_half(z, bufA, tw1, tw2, hl);
            // ac[2n] = Re Y[n], ac[2n+1] = -Im Y[n]; r[k] = ac[k] / (ac[0] windowR[k]) into the same region
            // (reciprocal to < 1 ulp and a multiply: the quotient is not correctly rounded, like the transforms before it)
            rr = reinterpret_cast<double *>(bufA);
            const double ac0 = __shfl(z[0].x, lane & ~(LW - 1), 64);
            if (active) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int k = 2 * (hl + LW * r);
                    if (k >= 1 && k <= P.bix) { const double v = z[r].x * rcp_f64(ac0 * winR[k]); rr[P.bix + k] = v; rr[P.bix - k] = v; }
                    if (k + 1 <= P.bix) { const double v = -z[r].y * rcp_f64(ac0 * winR[k + 1]); rr[P.bix + k + 1] = v; rr[P.bix - k - 1] = v; }
                }
                if (hl == 0) rr[P.bix] = 1.0;
            }
            wave_sync();
        } else {
        wave_sync();
        // forward transform of the packed frame
        double2 *Z = fft_wave(bufA, bufB, M, twM, lane);
        double2 *W = (Z == bufA) ? bufB : bufA;
        // power spectrum of the real frame, re-tangled for the second (inverse) transform
        for (int k = lane; k <= (M >> 1); k += 64) {
            const double2 zk = Z[ZP(k & (M - 1))], zm = Z[ZP((M - k) & (M - 1))];
            const double2 w = twN[k];                                     // (cos, -sin)
            const double2 ez = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
            const double2 oz = make_double2(0.5 * (zk.y + zm.y), -0.5 * (zk.x - zm.x));
            const double2 t = cmul_f64(oz, w);
            const double xr1 = ez.x + t.x, xi1 = ez.y + t.y, xr2 = ez.x - t.x, xi2 = ez.y - t.y;
            const double pk = fma(xr1, xr1, xi1 * xi1), pm = fma(xr2, xr2, xi2 * xi2);
            const double e = 0.5 * (pk + pm), d = 0.5 * (pk - pm);
            const double c = w.x, sn = -w.y;
            W[ZP(k & (M - 1))] = make_double2(e - d * sn, -(d * c));
            if (k != 0 && k != M - k) W[ZP(M - k)] = make_double2(e + d * sn, -(d * c));
        }
        wave_sync();
        double2 *Y = fft_wave(W, Z, M, twM, lane);
        rr = reinterpret_cast<double *>((Y == bufA) ? bufB : bufA);       // the free buffer holds r[-bix..bix]
        const double ac0 = Y[0].x;
        for (int k = lane + 1; k <= P.bix; k += 64) {
            const double2 y = Y[ZP(k >> 1)];
            const double ack = (k & 1) ? -y.y : y.x;
            const double v = ack / (ac0 * windowR[k]);
            rr[P.bix + k] = v; rr[P.bix - k] = v;
        }
        if (lane == 0) rr[P.bix] = 1.0;
        wave_sync();
        }

        const int ynx = 2 * P.bix + 1;
        const int lim = min(P.maxlag, P.bix);
        const double half_vt = 0.5 * P.voicing_thr;
        // count the local maxima first: with at most maxc-1 of them (the common case) every
        // maximum becomes a candidate and the first-pass strength is never consulted.
        int total = 0;
        for (int base = 2; base < lim; base += LW) {
            const int i = base + hl;
            bool pred = false;
            if (active && i < lim) {
                const double r0 = rr[P.bix + i], rm = rr[P.bix + i - 1], rp = rr[P.bix + i + 1];
                pred = r0 > half_vt && r0 > rm && r0 >= rp;
            }
            unsigned long long mask = __ballot(pred);
            if (LW == 32) mask = (mask >> (32 * half)) & 0xffffffffull;
            const int here = __popcll(mask);
            if (total + here <= P.maxc - 1) {
                // the lane that owns slot (1 + total + q) takes the lag of the q-th maximum of this round
                const int need = hl - 1 - total;
                if (need >= 0 && need < here) {
                    unsigned long long m = mask;
                    for (int q = 0; q < need; q++) m &= m - 1;
                    c_i = base + __ffsll((long long)m) - 1;
                }
            }
            total += here;
        }
        const bool rare = total > P.maxc - 1;
        if (!rare) n = 1 + total;
        // rare: more maxima than candidate slots -> Praat's replacement rule needs first-pass strengths.
        // The whole wavefront works on one frame at a time here (sinc_wave spreads its terms over 64 lanes).
        const unsigned long long rare_mask = __ballot(rare);
        for (int g = 0; g < FPW; g++) {
            if (!((rare_mask >> (g * LW)) & 1ull)) continue;
            const double *rg = FPW == 1 ? rr : reinterpret_cast<const double *>(wave_base + g * REG_C);
            const bool mine = half == g;
            int ng = 1;
            if (mine) c_i = 0;
            for (int base = 2; base < lim; base += 64) {
                const int i = base + lane;
                bool pred = false;
                if (i < lim) {
                    const double r0 = rg[P.bix + i], rm = rg[P.bix + i - 1], rp = rg[P.bix + i + 1];
                    pred = r0 > half_vt && r0 > rm && r0 >= rp;
                }
                unsigned long long mask = __ballot(pred);
                while (mask) {
                    const int bpos = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const int im = base + bpos;
                    const double r0 = rg[P.bix + im], rm = rg[P.bix + im - 1], rp = rg[P.bix + im + 1];
                    const double dr = 0.5 * (rp - rm), d2r = 2.0 * r0 - rm - rp;
                    const double fmx = 1.0 / P.dx / ((double)im + dr / d2r);
                    double smx = sinc_wave(rg, ynx, 1.0 / P.dx / fmx + (double)(P.bix + 1), 30, lane);
                    if (smx > 1.0) smx = 1.0 / smx;
                    int place = -1;
                    if (ng < P.maxc) {
                        place = ng++;
                    } else {
                        // weakest candidate so far among 1..maxc-1 (first minimum wins)
                        double ls = c_s - P.octave_cost * (log(P.min_pitch / c_f) * LOG2E_D);
                        int li = hl;
                        if (!mine || hl < 1 || hl >= P.maxc) { ls = 1e300; li = 1 << 20; }
                        for (int off = 32; off > 0; off >>= 1) {
                            const double os = __shfl_xor(ls, off, 64);
                            const int oi = __shfl_xor(li, off, 64);
                            if (os < ls || (os == ls && oi < li)) { ls = os; li = oi; }
                        }
                        double weakest = 2.0;
                        if (ls < weakest) { weakest = ls; place = li; }
                        if (smx - P.octave_cost * (log(P.min_pitch / fmx) * LOG2E_D) <= weakest) place = -1;
                    }
                    if (place >= 0 && mine && hl == place) { c_f = fmx; c_s = smx; c_i = im; }
                }
            }
            if (mine) n = ng;
        }
        if (n > 1) {
            // hand r[-bix..bix] and the candidate lags to k_pitch_refine
            // r is even: only r[0..bix] travels (half the bytes written here and read by k_pitch_refine)
            double *ro = rr_out + fidx * (int64_t)P.rr_half;
            for (int k = hl; k <= P.bix; k += LW) ro[k] = rr[P.bix + k];
            // append to one of RF_LISTS lists: a single counter would serialise ~10^5 returning atomics in L2
            const unsigned int list = (unsigned int)bid & (RF_LISTS - 1);
            unsigned int pos = 0;
            if (hl == 0) pos = atomicAdd(item_count + list * RF_CSTRIDE, (unsigned int)(n - 1));
            pos = __shfl(pos, lane & ~(LW - 1), 64);
            if (hl >= 1 && hl < n) items[(size_t)list * list_cap + pos + hl - 1] = RefineItem{(long long)fidx, hl, c_i};
        }
    }
    // (candidate slots are not cleared: slot 0 is the voiceless candidate by definition and slots >= n are
    //  never read -- k_pitch_delta and k_pitch_path go by ncand)
    if (live && hl == 0) { ncand[fidx] = n; intensity[fidx] = inten; }
    }   // frames of this wavefront
}

// ---------------------------------------------------------------------------
// k_pitch_refine: Praat's second pass (NUMimproveMaximum, sinc depth 70/700) over the flat
// list of candidates.  Eight lanes per candidate: the 144 autocorrelation values a
// depth-70 interpolation can touch live in registers (18 per lane, fixed absolute indices),
// the per-evaluation scalars (Brent state, sin) are shared by eight candidates per wave, and
// the 8-lane sums use DPP row operations (no LDS).  Trigonometry is evaluated with plain
// polynomials (every argument is known to lie in (0, pi]); along a lane's rows the raised-cosine
// arguments form an arithmetic progression, so only the two rows nearest x on each side take
// the polynomial and the rest follow by the three-term cosine recurrence.
// ---------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row; every lane of the row receives the same bits
__device__ __forceinline__ double row_sum16(double v)
{
    v += dpp_f64<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);    // row_half_mirror
    v += dpp_f64<0x140>(v);    // row_mirror
    return v;
}
// sum over the 8 lanes of half a DPP row
__device__ __forceinline__ double row_sum8(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    return v;
}
// cos(h) and sin(h) for h in [0, pi/2] (Taylor about 0; truncation < 1e-19), evaluated by Estrin's
// scheme: a dependent fp64 FMA costs ~30 cycles on this part (measured), so the 13-deep Horner chain
// was the critical path of every evaluation; this form is 5 deep for two more multiplies.
__device__ __forceinline__ double poly13_estrin(double z, double c0, double c1, double c2, double c3, double c4, double c5, double c6,
                                                double c7, double c8, double c9, double c10, double c11, double c12)
{
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double q0 = fma(c1, z, c0), q1 = fma(c3, z, c2), q2 = fma(c5, z, c4), q3 = fma(c7, z, c6), q4 = fma(c9, z, c8), q5 = fma(c11, z, c10);
    const double r0 = fma(q1, z2, q0), r1 = fma(q3, z2, q2), r2 = fma(q5, z2, q4);
    const double s0 = fma(r1, z4, r0), s1 = fma(c12, z4, r2);
    return fma(s1, z8, s0);
}
__device__ __forceinline__ double cos_q(double h)
{
    return poly13_estrin(h * h, 1.0, -0.5, 4.1666666666666664e-02, -1.3888888888888889e-03, 2.4801587301587302e-05, -2.7557319223985888e-07,
                         2.0876756987868099e-09, -1.1470745597729725e-11, 4.7794773323873853e-14, -1.5619206968586225e-16,
                         4.1103176233121648e-19, -8.8967913924505741e-22, 1.6117375710961184e-24);
}
__device__ __forceinline__ double sin_q(double h)
{
    // sin h = h + h z (s0 + s1 z + ... + s11 z^11), z = h^2
    const double z = h * h;
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double q0 = fma(8.3333333333333332e-03, z, -1.6666666666666666e-01), q1 = fma(2.7557319223985893e-06, z, -1.9841269841269841e-04);
    const double q2 = fma(1.6059043836821613e-10, z, -2.5052108385441720e-08), q3 = fma(2.8114572543455206e-15, z, -7.6471637318198164e-13);
    const double q4 = fma(1.9572941063391263e-20, z, -8.2206352466243295e-18), q5 = -3.8681701706306841e-23;
    const double r0 = fma(q1, z2, q0), r1 = fma(q3, z2, q2), r2 = fma(q5, z2, q4);
    const double p = fma(r2, z8, fma(r1, z4, r0));
    return fma(p * z, h, h);
}
__device__ __forceinline__ double sin_0pi(double a) { const double h = 0.5 * a; return 2.0 * sin_q(h) * cos_q(h); }
__device__ __forceinline__ double one_plus_cos_0pi(double a) { const double c = cos_q(0.5 * a); return 2.0 * c * c; }

// NUM_interpolate_sinc for a depth that stays inside the register window.
// yv[m] = y[wbase + l8 + 8 m] (1-based y index), m < 18; the caller guarantees 8 <= D and that
// midleft is ixmid-1 or ixmid with wbase = ixmid - 71, so window rows m <= 7 lie left of x,
// rows m >= 9 right of it and only row 8 straddles it.  Per lane the index distance k
// changes by 8 from row to row: the (-1)^k sign is one value per side.  sin(pi (1 - t)) is
// taken equal to sin(pi t) (Praat evaluates both; they differ by rounding only).
// The window factor 1 + cos(aa_k), aa_k = pi (k + frac) / (D + frac), is linear in k: with
// u_m = 1 + cos(aa0 + m delta), u_(m+1) = tc u_m - u_(m-1) + (2 - tc), tc = 2 cos(delta).
// Rows are walked outward from x, so a row inside the depth limit only ever depends on rows
// inside it (rows beyond the limit are discarded by the k < D select, whatever they hold).
__device__ __forceinline__ double sinc_w(double kd, double a0, double yu /* y (1 + cos) */)
{
    return yu * rcp_f64(fma(kd, PI_D, a0));         // the side's +-0.5 sin factor is applied once per side
}
template <int G> __device__ __forceinline__ double group_sum(double v)
{
    v += dpp_f64<0xB1>(v);                              // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);                              // quad_perm [2,3,0,1]
    if (G == 8) v += dpp_f64<0x141>(v);                 // row_half_mirror
    return v;
}
// G lanes per candidate, NR = 144 / G rows per lane; row RS straddles x, rows < RS lie left, rows > RS right.
template <int G>
__device__ __forceinline__ double sinc_group_reg(const double (&yv)[144 / G], int wbase, int ynx, double x, int maxDepth, int lg)
{
    constexpr int NR = 144 / G, RS = (72 - G) / G;
    const int midleft = (int)floor(x), midright = midleft + 1;
    if (x == (double)midleft) {
        double pick = 0.0;
#pragma unroll
        for (int m = 0; m < NR; m++) if (wbase + lg + G * m == midleft) pick = yv[m];
        return group_sum<G>(pick);
    }
    int D = maxDepth;
    if (D > midright - 1) D = midright - 1;
    if (D > ynx - midleft) D = ynx - midleft;
    const double dlim = (double)D;
    const int left = midright - D, right = midleft + D;
    const double a_l = PI_D * (x - (double)midleft), a_r = PI_D * ((double)midright - x);
    // sin(pi f) = sin(pi (1 - f)): fold to [0, pi/2] and take ONE polynomial (sin_0pi costs a sine and a cosine)
    const double hs = 0.5 * sin_q(fmin(a_l, a_r));
    const double rden_l = rcp_f64(x - (double)left + 1.0), rden_r = rcp_f64((double)right - x + 1.0);
    const double aa_l = a_l * rden_l, daa_l = PI_D * rden_l;
    const double aa_r = a_r * rden_r, daa_r = PI_D * rden_r;
    const int kl0 = midleft - wbase - lg;             // k of row 0 on the left side (decreases by G per row)
    const int kr0 = wbase + lg - midright;            // k of row 0 on the right side (increases by G per row)
    const double hs_l = (kl0 & 1) ? -hs : hs, hs_r = (kr0 & 1) ? -hs : hs;
    const double kdl = (double)kl0, kdr = (double)kr0;
    // rows < RS and > RS have k >= 0 by construction (wbase = ixmid - 71): only the depth limit can drop them
    auto keep = [&](double kd, double t) { return kd < dlim ? t : 0.0; };
    double acc;
    {   // row RS holds the lanes around x
        const bool is_left = kl0 - G * RS >= 0;
        const double kd = is_left ? kdl - (double)(G * RS) : kdr + (double)(G * RS);
        const double u = one_plus_cos_0pi(fma(kd, is_left ? daa_l : daa_r, is_left ? aa_l : aa_r));
        const double ts = (is_left ? hs_l : hs_r) * sinc_w(kd, is_left ? a_l : a_r, yv[RS] * u);
        acc = (kd >= 0.0 && kd < dlim) ? ts : 0.0;
    }
    // Two rows share one reciprocal: t0/a0 + t1/a1 = (t0 a1 + t1 a0) / (a0 a1) (a = pi (k + frac) > 0; a v_rcp_f64
    // issues at quarter rate), so the rows are first collected (masked numerators t = y (1 + cos), denominators a)
    // and then folded in pairs.
    {   // left side, rows RS-1 (nearest x) .. 0
        constexpr int NL = RS;
        double tv[NL], av[NL];
        const double cq = cos_q(0.5 * G * daa_l);                       // cos(delta / 2), delta = G daa <= pi
        const double tc = fma(4.0 * cq, cq, -2.0), g = 2.0 - tc;
        const double k1 = kdl - (double)(G * (RS - 1)), k0 = kdl - (double)(G * (RS - 2));
        double u1 = one_plus_cos_0pi(fma(k1, daa_l, aa_l));
        double u0 = one_plus_cos_0pi(fma(k0, daa_l, aa_l));
        tv[0] = keep(k1, yv[RS - 1] * u1); av[0] = fma(k1, PI_D, a_l);
        tv[1] = keep(k0, yv[RS - 2] * u0); av[1] = fma(k0, PI_D, a_l);
#pragma unroll
        for (int m = RS - 3; m >= 0; m--) {
            const double u = fma(tc, u0, g - u1);        // (g - u1) is off the critical path
            u1 = u0; u0 = u;
            const double kd = kdl - (double)(G * m);
            tv[RS - 1 - m] = keep(kd, yv[m] * u); av[RS - 1 - m] = fma(kd, PI_D, a_l);
        }
        double side = 0.0;
#pragma unroll
        for (int q = 0; q + 1 < NL; q += 2) side += fma(tv[q], av[q + 1], tv[q + 1] * av[q]) * rcp_f64(av[q] * av[q + 1]);
        if (NL & 1) side += tv[NL - 1] * rcp_f64(av[NL - 1]);
        acc = fma(hs_l, side, acc);
    }
    {   // right side, rows RS+1 (nearest x) .. NR-1
        constexpr int NRt = NR - RS - 1;
        double tv[NRt], av[NRt];
        const double cq = cos_q(0.5 * G * daa_r);
        const double tc = fma(4.0 * cq, cq, -2.0), g = 2.0 - tc;
        const double k1 = kdr + (double)(G * (RS + 1)), k0 = kdr + (double)(G * (RS + 2));
        double u1 = one_plus_cos_0pi(fma(k1, daa_r, aa_r));
        double u0 = one_plus_cos_0pi(fma(k0, daa_r, aa_r));
        tv[0] = keep(k1, yv[RS + 1] * u1); av[0] = fma(k1, PI_D, a_r);
        tv[1] = keep(k0, yv[RS + 2] * u0); av[1] = fma(k0, PI_D, a_r);
#pragma unroll
        for (int m = RS + 3; m < NR; m++) {
            const double u = fma(tc, u0, g - u1);        // (g - u1) is off the critical path
            u1 = u0; u0 = u;
            const double kd = kdr + (double)(G * m);
            tv[m - RS - 1] = keep(kd, yv[m] * u); av[m - RS - 1] = fma(kd, PI_D, a_r);
        }
        double side = 0.0;
#pragma unroll
        for (int q = 0; q + 1 < NRt; q += 2) side += fma(tv[q], av[q + 1], tv[q + 1] * av[q]) * rcp_f64(av[q] * av[q + 1]);
        if (NRt & 1) side += tv[NRt - 1] * rcp_f64(av[NRt - 1]);
        acc = fma(hs_r, side, acc);
    }
    return group_sum<G>(acc);
}

// generic NUM_interpolate_sinc with y in global memory (depth 700, or windows near the array ends)
template <int G> __device__ double sinc_group_mem(const double *__restrict__ yh /* r[0..bix] */, int ynx, double x, int maxDepth, int lg)
{
    const int mid = (ynx + 1) >> 1;                      // Praat's y(i) = r[i - mid] = yh[|i - mid|]
    auto Y = [&](int i) { return yh[abs(i - mid)]; };
    const int midleft = (int)floor(x), midright = midleft + 1;
    if (x > (double)ynx) return Y(ynx);
    if (x < 1.0) return Y(1);
    if (x == (double)midleft) return Y(midleft);
    if (maxDepth > midright - 1) maxDepth = midright - 1;
    if (maxDepth > ynx - midleft) maxDepth = ynx - midleft;
    if (maxDepth <= 0) return Y((int)floor(x + 0.5));
    if (maxDepth == 1) return Y(midleft) + (x - (double)midleft) * (Y(midright) - Y(midleft));
    if (maxDepth == 2) {
        const double yl = Y(midleft), yr = Y(midright);
        const double dyl = 0.5 * (yr - Y(midleft - 1)), dyr = 0.5 * (Y(midright + 1) - yl);
        const double fil = x - (double)midleft, fir = (double)midright - x;
        return yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2.0 * (yr - yl)));
    }
    const int left = midright - maxDepth, right = midleft + maxDepth;
    const double a_l = PI_D * (x - (double)midleft), a_r = PI_D * ((double)midright - x);
    const double hs_l = 0.5 * sin_0pi(a_l), hs_r = 0.5 * sin_0pi(a_r);
    const double den_l = x - (double)left + 1.0, den_r = (double)right - x + 1.0;
    const double aa_l = a_l / den_l, daa_l = PI_D / den_l;
    const double aa_r = a_r / den_r, daa_r = PI_D / den_r;
    double acc = 0.0;
    for (int t = lg; t < 2 * maxDepth; t += G) {
        const bool is_left = t < maxDepth;
        const int k = is_left ? t : t - maxDepth;
        const double kd = (double)k;
        const double a = fma(kd, PI_D, is_left ? a_l : a_r);
        const double aa = fma(kd, is_left ? daa_l : daa_r, is_left ? aa_l : aa_r);
        double hs = is_left ? hs_l : hs_r;
        if (k & 1) hs = -hs;
        const int ix = is_left ? midleft - k : midright + k;
        acc += Y(ix) * (hs * rcp_f64(a) * one_plus_cos_0pi(aa));
    }
    return group_sum<G>(acc);
}

template <int G>
__global__ __launch_bounds__(256, G == 8 ? 3 : 2) void k_pitch_refine(PiParams P, const double *__restrict__ rr_in, const RefineItem *__restrict__ items,
                                                     const unsigned int *__restrict__ item_count, unsigned int list_cap,
                                                     double *__restrict__ cand)
{
    constexpr int NR = 144 / G, GS = G == 8 ? 3 : 2;
    const int l8 = threadIdx.x & (G - 1);
    const unsigned int gid = (blockIdx.x * blockDim.x + threadIdx.x) >> GS;
    const unsigned int list = gid & (RF_LISTS - 1);
    const unsigned int group = gid / RF_LISTS;
    const unsigned int n_groups = ((gridDim.x * blockDim.x) >> GS) / RF_LISTS;
    const unsigned int count = item_count[list * RF_CSTRIDE];
    items += (size_t)list * list_cap;
    const int ynx = 2 * P.bix + 1;
    const double golden = 1.0 - 0.6180339887498948482045868343656381177203;
    const double tol = 1e-10;
    auto finish = [&](const RefineItem &item, double xres, double yres) {
        xres -= (double)(P.bix + 1);
        if (yres > 1.0) yres = 1.0 / yres;
        if (l8 == 0) {
            cand[item.frame * 32 + item.slot] = 1.0 / P.dx / xres;
            cand[item.frame * 32 + 16 + item.slot] = yres;
        }
    };
    // The maximiser as a state machine: every trip of the loop is ONE function evaluation for each of the wave's eight candidates, and a
    // group whose candidate has converged fetches its next item at once, so candidates with different iteration counts do not wait for
    // one another.  Two searches (round 3):
    //   * successive parabolic interpolation seeded with the three samples around the maximum (their function values are the
    //     autocorrelation samples themselves: no evaluation): the first trial point already is the parabolic estimate, and the fit through
    //     the best three points converges superlinearly: 3-4 evaluations (stop: a step below 3e-8 |x| after a step below 1e-3).
    //     Praat's NUMminimize_brent (a golden-section opening of [ixmid - 1, ixmid + 1], then a bracket that has to close to 1.5e-8 |x|
    //     from both sides) takes 9-18 for the same maximum, whose position it reports within that tolerance: 1e-6 lags absolute, 1.5e-8
    //     relative in F0, against a parity gate of 1e-6;
    //   * Praat's own iterates, exactly as before, for the candidates where the two could differ by more than rounding: the
    //     interpolant is smooth on either side of the sample ixmid but has a kink AT it (the sinc window changes with floor(x)), so a
    //     maximum within a few thousandths of a lag of the sample can split into one local maximum per side (measured on synthetic
    //     autocorrelations: discrepancies of 2e-5 .. 7e-5 relative, all within 0.007 lags of the sample) and which one the reference
    //     reports depends on its path.  Every candidate whose parabolic search ends within 0.03 lags of the sample (3 %), or fails a
    //     safeguard (step outside the bracket, no convergence in 10 steps), is searched again the reference's way.
    // Elsewhere the maximum in the bracket is unique, both searches end within their tolerance of it, and the strengths agree to second
    // order (the maximum is flat).  PCE_PITCH_REFINE=praat runs every candidate the reference's way.
    unsigned int it = group;
    bool have = false;
    RefineItem item = {0, 0, 0};
    const double *y = rr_in;
    int wbase = 0, depth = 70, iter = 0;             // iter >= 0: Brent's iteration count; iter < 0: parabolic search, -1 - evaluations so far
    bool fast = false;
    double yv[NR];
    double a = 0.0, b = 0.0, v = 0.0, w = 0.0, x = 0.0, fv = 0.0, fw = 0.0, fx = 0.0, t = 0.0, prev_step = 1.0;
    const double sqrt_epsilon = 1.4901161193847656e-08, spi_stop = P.refine_tol_rel;
    // Brent's choice of the next trial point from the state (a, b, x, w, v); true: converged
    auto brent_next = [&]() -> bool {
        const double range = b - a;
        const double middle_range = (a + b) / 2.0;
        const double tol_act = sqrt_epsilon * fabs(x) + tol / 3.0;
        if (fabs(x - middle_range) + range / 2.0 <= 2.0 * tol_act) return true;
        double new_step = golden * (x < middle_range ? b - x : a - x);
        if (fabs(x - w) >= tol_act) {
            double tt = (x - w) * (fx - fv);
            double q = (x - v) * (fx - fw);
            double p = (x - v) * q - (x - w) * tt;
            q = 2.0 * (q - tt);
            if (q > 0.0) p = -p; else q = -q;
            if (fabs(p) < fabs(new_step * q) && p > q * (a - x + 2.0 * tol_act) && p < q * (b - x - 2.0 * tol_act))
                new_step = p * rcp_f64(q);          // (an IEEE divide is ~35 instructions; this is within 1 ulp of it)
        }
        if (fabs(new_step) < tol_act) new_step = new_step > 0.0 ? tol_act : -tol_act;
        t = x + new_step;
        return false;
    };
    auto brent_start = [&](double node) {
        a = node - 1.0; b = node + 1.0;
        v = a + golden * (b - a);
        t = v; iter = 0;
    };
    // next point of the parabolic search: 0 = evaluate t, 1 = converged at x, 2 = leave it to Brent
    auto spi_next = [&]() -> int {
        const double tt = (x - w) * (fx - fv);
        double q = (x - v) * (fx - fw);
        double p = (x - v) * q - (x - w) * tt;
        q = 2.0 * (q - tt);
        if (q == 0.0) return 2;
        if (q > 0.0) p = -p; else q = -q;
        const double step = p * rcp_f64(q), tn = x + step;
        if (!(tn > a && tn < b)) return 2;
        const int n_eval = -1 - iter;
        if (n_eval >= 2 && fabs(step) <= spi_stop * fabs(x) && fabs(prev_step) <= 1e-3) return 1;
        if (n_eval >= 10) return 2;
        prev_step = step; t = tn;
        return 0;
    };
    // a group's next item is fetched one item ahead and the 21 autocorrelation values of an item are requested together: one memory
    // round trip per item stands in the wave's instruction stream, not three
    RefineItem nxt = {0, 0, 0};
    if (it < count) nxt = items[it];
    for (;;) {
        while (!have && it < count) {
            item = nxt;
            it += n_groups;
            if (it < count) nxt = items[it];
            y = rr_in + item.frame * (long long)P.rr_half;               // Praat's y(i) = r[i - mid] = y[|i - mid|], mid = bix + 1
            const int ixmid = item.imax + P.bix + 1;
            if (ixmid <= 1) { finish(item, 1.0, y[P.bix]); continue; }
            if (ixmid >= ynx) { finish(item, (double)ynx, y[P.bix]); continue; }
            // depth choice uses the first-pass (parabolic) frequency, as Praat does
            const double r0 = y[abs(item.imax)], rm = y[abs(item.imax - 1)], rp = y[abs(item.imax + 1)];
            const double dr = 0.5 * (rp - rm), d2r = 2.0 * r0 - rm - rp;
            const double f1 = 1.0 / P.dx / ((double)item.imax + dr / d2r);
            depth = f1 > 0.3 / P.dx ? 700 : 70;
            // the register window serves midleft in {ixmid-1, ixmid}: needs depth <= 70 and D >= 8 for both
            fast = depth == 70 && ixmid - 1 >= 9 && ynx - ixmid >= 8;
            wbase = ixmid - 71;
#pragma unroll
            for (int m = 0; m < NR; m++) {                                 // (loaded whether or not the fast path will use them)
                const int ix = wbase + l8 + G * m;
                yv[m] = (ix >= 1 && ix <= ynx) ? y[abs(ix - P.bix - 1)] : 0.0;
            }
            have = true;
            if (P.refine_seeded) {
                // the interpolant passes through the samples: f(ixmid) = -r0, f(ixmid -+ 1) = -rm / -rp
                a = (double)(ixmid - 1); b = (double)(ixmid + 1);
                x = (double)ixmid; fx = -r0;
                if (rm >= rp) { w = a; fw = -rm; v = b; fv = -rp; } else { w = b; fw = -rp; v = a; fv = -rm; }
                iter = -1; prev_step = 1.0;
                if (spi_next() != 0) brent_start((double)ixmid);
            } else brent_start((double)ixmid);
        }
        if (__ballot(have) == 0) break;
        if (have) {
            const double ft = fast ? -sinc_group_reg<G>(yv, wbase, ynx, t, 70, l8) : -sinc_group_mem<G>(y, ynx, t, depth, l8);
            if (iter == 0) {
                x = v; w = v; fx = ft; fw = ft; fv = ft;
            } else if (ft <= fx) {
                if (t < x) b = x; else a = x;
                v = w; w = x; x = t;
                fv = fw; fw = fx; fx = ft;
            } else {
                if (t < x) a = t; else b = t;
                if (ft <= fw || w == x) { v = w; w = t; fv = fw; fw = ft; }
                else if (ft <= fv || v == x || v == w) { v = t; fv = ft; }
            }
            if (iter >= 0) {
                iter++;
                if (iter > 60 || brent_next()) { finish(item, x, -fx); have = false; }
            } else {
                iter--;
                const double node = (double)(item.imax + P.bix + 1);
                const int r = spi_next();
                if (r == 1 && fabs(x - node) >= 0.03) { finish(item, x, -fx); have = false; }
                else if (r != 0) brent_start(node);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Pitch_pathFinder: one wavefront per slice
// ---------------------------------------------------------------------------
constexpr int BT_TILE = 256;              // frames per back-tracking tile
constexpr int RUN_LISTS = 64;             // independent run lists (see RF_LISTS)
constexpr double PATH_VOICELESS = 1e300;

struct PathRun { long long frame; long long gend; int has_prev; int pad; };   // run start, slice end (global frame indices), a cut frame precedes

__device__ __forceinline__ double readlane_f64(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Elementwise pre-pass of Pitch_pathFinder over every (frame, candidate):
//   dl = {local delta (the finder's first loop), log2 f or PATH_VOICELESS}
// plus the bookkeeping that makes the Viterbi parallel: a frame with a single candidate (the
// voiceless one) is a *cut*: every path passes through it, and the additive constant it
// carries cannot change any later arg-max.  So the recurrence only has to run over maximal
// runs of multi-candidate frames, all runs independently.  Cut frames get their result
// (f0 = 0) here; run starts are appended to RUN_LISTS lists.
__global__ __launch_bounds__(256) void k_pitch_delta(PiParams P, const PiSlice *__restrict__ slices, const int *__restrict__ frame_slice,
                                                    const double *__restrict__ cand, const int *__restrict__ ncand,
                                                    const double *__restrict__ intensity, long long n_frames,
                                                    double2 *__restrict__ dl /* [frames][16] */, double *__restrict__ f0,
                                                    double *__restrict__ strength, PathRun *__restrict__ runs,
                                                    unsigned int *__restrict__ run_count, unsigned int run_cap)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_frames * PI_MAXC) return;
    const long long gi = e >> 4; const int jc = (int)(e & 15);
    const int n_here = ncand[gi];
    const bool real = jc >= 1 && jc < n_here;             // slot 0 is the voiceless candidate; slots >= n do not exist
    const double f = real ? cand[gi * 32 + jc] : 0.0, st = real ? cand[gi * 32 + 16 + jc] : 0.0;
    const double inten = intensity[gi];
    double uv = P.silence_thr <= 0.0 ? 0.0 : 2.0 - inten / (P.silence_thr / (1.0 + P.voicing_thr));
    uv = P.voicing_thr + (uv > 0.0 ? uv : 0.0);
    const bool voiceless_local = f == 0.0 || f > P.ceiling;
    const bool voiceless_trans = f <= 0.0 || f >= P.ceiling;
    const double delta = voiceless_local ? uv : st - P.octave_cost * (log(P.ceiling / f) * LOG2E_D);
    if (jc < max(n_here, 1)) dl[e] = make_double2(delta, voiceless_trans ? PATH_VOICELESS : log(f) * LOG2E_D);
    if (jc == 0) {
        const int n = n_here;
        if (n <= 1) { f0[gi] = 0.0; strength[gi] = 0.0; }
        else {
            const int sl = frame_slice[gi];
            const long long li = gi - slices[sl].frame_off;
            if (li == 0 || ncand[gi - 1] <= 1) {
                const unsigned int list = (unsigned int)(gi >> 2) & (RUN_LISTS - 1);
                const unsigned int pos = atomicAdd(run_count + list * RF_CSTRIDE, 1u);
                runs[(size_t)list * run_cap + pos] = PathRun{gi, slices[sl].frame_off + slices[sl].n_frames, li > 0 ? 1 : 0, 0};
            }
        }
    }
}

// Viterbi over one run of multi-candidate frames per wavefront.  Lane ic (0..15) owns
// candidate ic of the current frame; the previous frame's running delta and log2-frequency
// stay in those lanes' registers and are broadcast with v_readlane (the loop over previous
// candidates is wave-uniform), so the recurrence touches neither a barrier nor global memory.
// Runs are short (speech: tens of frames, a few hundred at most), so everything around the
// recurrence is sized for that: operands arrive in 16-frame chunks (one 16-byte load per lane
// and four frames per instruction, the next chunk in flight while this one is consumed), the
// back-pointers stay in LDS for runs up to PT_CAP frames, and the workgroup is the wavefront
// (no __syncthreads).  Longer runs back-track through global memory in tiles.
constexpr int PT_CHUNK = 16;
constexpr int PT_CAP = 1024;
__global__ __launch_bounds__(64) void k_pitch_path(
    PiParams P, const double *__restrict__ cand, const int *__restrict__ ncand,
    const double2 *__restrict__ dl, const PathRun *__restrict__ runs, const unsigned int *__restrict__ run_count, unsigned int run_cap,
    unsigned char *__restrict__ psi /* [frames][16] */, double *__restrict__ f0, double *__restrict__ strength)
{
    __shared__ double2 t_dl[2][PT_CHUNK][PI_MAXC];
    __shared__ int t_n[2][PT_CHUNK];
    __shared__ __attribute__((aligned(16))) unsigned char t_psi[PT_CAP][PI_MAXC];
    __shared__ unsigned char t_place[PT_CAP];
    static_assert(BT_TILE <= PT_CAP, "the global back-tracking tiles reuse t_psi");
    const int lane = threadIdx.x;
    const int ic = lane & 15;
    const double timeStepCorrection = 0.01 / P.dt;
    const double ojc = P.oj_cost * timeStepCorrection, vuc = P.vuv_cost * timeStepCorrection;
    const unsigned int list = blockIdx.x & (RUN_LISTS - 1);
    const unsigned int count = run_count[list * RF_CSTRIDE];
    for (unsigned int r = blockIdx.x / RUN_LISTS; r < count; r += gridDim.x / RUN_LISTS) {
        const PathRun run = runs[(size_t)list * run_cap + r];
        const long long gs = run.frame, gend = run.gend;
        double pd = 0.0, plf = PATH_VOICELESS;           // previous frame, candidate `ic`
        int pn = 0;
        if (run.has_prev) { pd = dl[(gs - 1) * PI_MAXC].x; pn = 1; }   // the cut frame before the run: one voiceless candidate
        double2 rg[4]; int rn = 0;                        // chunk in flight
        auto fetch = [&](long long g0) {
            const int tn = (int)min((long long)PT_CHUNK, gend - g0);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int e = lane + 64 * q;
                rg[q] = (e >> 4) < tn ? dl[g0 * PI_MAXC + e] : make_double2(0.0, 0.0);
            }
            rn = lane < tn ? ncand[g0 + lane] : 0;        // 0 candidates past the slice end: the run stops there
        };
        auto stage = [&](int b) {
#pragma unroll
            for (int q = 0; q < 4; q++) (&t_dl[b][0][0])[lane + 64 * q] = rg[q];
            if (lane < PT_CHUNK) t_n[b][lane] = rn;
        };
        wave_sync();                                      // the previous run's LDS readers are done
        fetch(gs);
        stage(0);
        long long ge = gs;                                // one past the last frame of the run
        int buf = 0;
        bool open = true;
        while (open) {
            const bool more = ge + PT_CHUNK < gend;
            if (more) fetch(ge + PT_CHUNK);
            wave_sync();
            int fr = 0;
            int n2 = t_n[buf][0];
            double2 cur = t_dl[buf][0][ic];
            for (; fr < PT_CHUNK; fr++) {
                if (n2 <= 1) { open = false; break; }
                const int nx = fr + 1 < PT_CHUNK ? fr + 1 : fr;
                const int n2n = t_n[buf][nx];                 // next frame's operands: off the recurrence's critical path
                const double2 nxt = t_dl[buf][nx][ic];
                const double d2 = cur.x, lf2 = cur.y;
                double best = d2; int place = 0;
                if (pn > 0) {
                    best = -1e30;
                    const bool cur_vl = lf2 > 1e299;
                    for (int ic1 = 0; ic1 < pn; ic1++) {
                        const double lf1 = readlane_f64(plf, ic1), d1 = readlane_f64(pd, ic1);
                        const bool pv = lf1 > 1e299;
                        const double tc = (cur_vl != pv) ? vuc : (cur_vl ? 0.0 : ojc * fabs(lf1 - lf2));
                        const double value = d1 - tc + d2;
                        const bool better = value > best;
                        best = better ? value : best; place = better ? ic1 : place;
                    }
                }
                pd = best; plf = lf2; pn = n2;
                const long long idx = ge + fr - gs;
                if (lane < PI_MAXC) {
                    psi[(ge + fr) * PI_MAXC + ic] = (unsigned char)place;
                    if (idx < PT_CAP) t_psi[idx][ic] = (unsigned char)place;
                }
                cur = nxt; n2 = n2n;
            }
            ge += fr;
            if (open) {
                if (more) { stage(buf ^ 1); buf ^= 1; }
                else open = false;
            }
        }
        // choose the end of the path inside the run
        int place = 0;
        {
            double maximum = -1e30;
            const bool cut_follows = ge < gend;          // frame `ge` is a cut frame (single voiceless candidate)
            const double dn = cut_follows ? dl[ge * PI_MAXC].x : 0.0;
            for (int jc = 0; jc < pn; jc++) {
                const double v = readlane_f64(pd, jc), lf = readlane_f64(plf, jc);
                const double value = cut_follows ? (v - ((lf > 1e299) ? 0.0 : vuc) + dn) : v;
                if (jc == 0 || value > maximum) { place = jc; maximum = value; }
            }
        }
        const long long L = ge - gs;
        wave_sync();
        if (L <= PT_CAP) {
            // back-track in LDS (one lane walks the chain), then every lane emits frames
            if (lane == 0) {
                int pl = place;
                for (int i = (int)L - 1; i >= 0; i--) { t_place[i] = (unsigned char)pl; pl = t_psi[i][pl]; }
            }
            wave_sync();
            for (int e = lane; e < (int)L; e += 64) {
                const long long gi = gs + e;
                const int pl = t_place[e];
                f0[gi] = pl ? cand[gi * 32 + pl] : 0.0;
                strength[gi] = pl ? cand[gi * 32 + 16 + pl] : 0.0;
            }
        } else {
            // long run: back-track through the global back-pointers in tiles, last tile first
            __threadfence();
            for (long long hi = ge; hi > gs;) {
                const long long lo = max(gs, hi - BT_TILE);
                const int cnt = (int)(hi - lo);
                for (int e = lane; e < cnt; e += 64)
                    *reinterpret_cast<uint4 *>(&t_psi[e][0]) = *reinterpret_cast<const uint4 *>(psi + (lo + e) * PI_MAXC);
                wave_sync();
                if (lane == 0) {
                    int pl = place;
                    for (int i = cnt - 1; i >= 0; i--) { t_place[i] = (unsigned char)pl; pl = t_psi[i][pl]; }
                    place = pl;
                }
                place = __shfl(place, 0, 64);
                wave_sync();
                for (int e = lane; e < cnt; e += 64) {
                    const long long gi = lo + e;
                    const int pl = t_place[e];
                    f0[gi] = pl ? cand[gi * 32 + pl] : 0.0;
                    strength[gi] = pl ? cand[gi * 32 + 16 + pl] : 0.0;
                }
                wave_sync();
                hi = lo;
            }
        }
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
    if (s.n_frames > npow2) return;                       // longer than the LDS sort holds: k_pitch_median_long's
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

// The same summary for a slice of any length (get_median_pitch takes a whole recording, Code/audioPipeline.py:326-335: the reference has no
// limit): the k-th smallest voiced F0 by radix selection on the bit patterns (positive doubles order as their 64-bit integers), eight
// 256-bin passes over the slice's F0 track in global memory per selected rank.  Exact: the median is np.median's (the middle element, or
// the mean of the two middle ones); the log sum adds in the order k_pitch_median adds (per thread by stride 256, then the same tree).
// One workgroup per slice; slices the LDS sort holds return at once.
__global__ __launch_bounds__(256) void k_pitch_median_long(const PiSlice *__restrict__ slices, const double *__restrict__ f0, int lds_limit,
                                                          PiSummaryDev *__restrict__ out)
{
    __shared__ int hist[256];
    __shared__ int s_cnt;
    __shared__ double s_red[4];
    __shared__ unsigned long long s_prefix;
    __shared__ long long s_k;
    const PiSlice s = slices[blockIdx.x];
    const int tid = threadIdx.x;
    if (s.status != PCE_SLICE_OK || s.n_frames <= lds_limit) return;
    const double *f = f0 + s.frame_off;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    double lsum = 0.0; int mine = 0;
    for (int i = tid; i < s.n_frames; i += 256) {
        const double v = f[i];
        if (v > 0.0) { mine++; lsum += log(v); }
    }
    atomicAdd(&s_cnt, mine);
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = lsum;
    __syncthreads();
    const int nv = s_cnt;
    double sel[2] = {0.0, 0.0};
    const int want = nv == 0 ? 0 : ((nv & 1) ? 1 : 2);
    for (int q = 0; q < want; q++) {
        if (tid == 0) { s_prefix = 0ull; s_k = (nv & 1) ? nv / 2 : nv / 2 - 1 + q; }
        __syncthreads();
        for (int shift = 56; shift >= 0; shift -= 8) {
            hist[tid] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            for (int i = tid; i < s.n_frames; i += 256) {
                const double v = f[i];
                if (v > 0.0) {
                    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
                    if (shift == 56 || (b >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(int)((b >> shift) & 255ull)], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                long long k = s_k; int bin = 0;
                while (bin < 255 && k >= hist[bin]) { k -= hist[bin]; bin++; }
                s_k = k; s_prefix = prefix | ((unsigned long long)bin << shift);
            }
            __syncthreads();
        }
        sel[q] = __longlong_as_double((long long)s_prefix);
        __syncthreads();
    }
    if (tid == 0) {
        PiSummaryDev o;
        o.n_voiced = nv;
        if (nv == 0) { o.median = 0.0; o.mean_log = 0.0; }
        else {
            o.median = (nv & 1) ? sel[0] : (sel[0] + sel[1]) / 2.0;
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

int pce_pitch_set_refine(pce_ctx *c, int32_t mode)
{
    if (!c || (mode != PCE_REFINE_SEEDED && mode != PCE_REFINE_PRAAT)) return PCE_E_INVALID;
    const bool praat = mode == PCE_REFINE_PRAAT;
    if (c->pitch_refine_praat != praat) { c->pitch_refine_praat = praat; c->pi_cache.drop(); c->pi_params_valid = false; }
    return PCE_OK;
}

int pce_pitch_run(pce_ctx *c, const pce_pitch_params *p, const pce_slice *slices, int32_t n)
{
    if (!c || !p || (!slices && n > 0) || n < 0) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    PCE_HIP(c, hipSetDevice(c->device));
    { int rc = pce_side_join(c, pce_ctx::SIDE_TAIL); if (rc) return rc; }   // the previous run's tail still owns the pitch buffers
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
            int nfft = 8;
            while ((double)nfft < (double)P.nw * 1.5) nfft *= 2;          // Praat's nsampFFT
            P.nfft = nfft;
            const int Mc = nfft / 2;
            P.zlen = Mc + Mc / 8 + 2;
            P.rr_len = 2 * P.bix + 2;
            P.rr_half = (P.bix + 2) & ~1;                  // handoff row: r[0..bix], even length
            P.refine_seeded = c->pitch_refine_praat ? 0 : 1;
            P.refine_tol_rel = 3e-8;
            const size_t lds_wave = sizeof(double) * 4 * (size_t)P.zlen;            // two complex buffers per wavefront
            if (lds_wave > 160 * 1024) return pce_fail(c, PCE_E_LIMIT, "analysis window of %d samples does not fit LDS", P.nw);
            // register-resident transforms where N allows it and r[-bix..bix] fits the exchange region
            // (tables stay in global memory: staging them in LDS per workgroup measured slower than L1 hits)
            P.mode = (nfft == 1024 && P.rr_len <= R_WAVE_F64) ? 1
                     : (nfft == 512 && P.rr_len <= R_WAVE_F64 / 2) ? 2
                     : (nfft == 2048 && P.rr_len <= R3_WAVE_F64) ? 3 : 0;
            P.fpb = P.mode == 2 ? 2 * PI_FPB : PI_FPB;
            P.tabs = P.mode == 2;   // MODE 3: 75 KB of exchange images leave no room
            std::vector<double> tw((size_t)(Mc + Mc + 1) * 2);
            for (int m = 0; m < Mc; m++) { tw[2 * (size_t)m] = std::cos(2.0 * PI_D * m / Mc); tw[2 * (size_t)m + 1] = -std::sin(2.0 * PI_D * m / Mc); }
            for (int k = 0; k <= Mc; k++) { tw[2 * (size_t)(Mc + k)] = std::cos(2.0 * PI_D * k / nfft); tw[2 * (size_t)(Mc + k) + 1] = -std::sin(2.0 * PI_D * k / nfft); }
            PCE_HIP(c, c->pi_tw.reserve(sizeof(double) * tw.size()));
            PCE_HIP(c, hipMemcpyAsync(c->pi_tw.p, tw.data(), sizeof(double) * tw.size(), hipMemcpyHostToDevice, c->stream));
            std::vector<double> window, windowR;
            make_window_tables(pl, window, windowR);
            PCE_HIP(c, c->pi_window.reserve(sizeof(double) * window.size()));
            PCE_HIP(c, c->pi_windowR.reserve(sizeof(double) * windowR.size()));
            PCE_HIP(c, hipMemcpyAsync(c->pi_window.p, window.data(), sizeof(double) * window.size(), hipMemcpyHostToDevice, c->stream));
            PCE_HIP(c, hipMemcpyAsync(c->pi_windowR.p, windowR.data(), sizeof(double) * windowR.size(), hipMemcpyHostToDevice, c->stream));
            P.pcm_span = (((int)std::ceil((double)(P.fpb - 1) * P.dt / P.dx) + P.nw + 4) + 7) & ~7;   // samples under one work item's frames
            std::vector<double> blob;
            if (P.mode != 0) {
                // lane-ordered tables for the register paths: tw1[k-1][l] = W_M^(l k), tw2[k-1][c] = W_M^(8 c k),
                // twN[k] = W_N^k (k < M), window, windowR[0..bix]; every table starts on a 16-byte boundary
                const int LW = P.mode == 2 ? 32 : 64, CW = LW / 8, NK1 = P.mode == 3 ? 16 : 8, S2 = P.mode == 3 ? 16 : 8;
                auto W = [&](int m, int period) { return std::pair<double, double>(std::cos(2.0 * PI_D * m / period), -std::sin(2.0 * PI_D * m / period)); };
                for (int k = 1; k < NK1; k++) for (int l = 0; l < LW; l++) { auto w = W(l * k, Mc); blob.push_back(w.first); blob.push_back(w.second); }
                P.o_tw2 = (int)blob.size();
                for (int k = 1; k < 8; k++) for (int q = 0; q < CW; q++) { auto w = W(S2 * q * k, Mc); blob.push_back(w.first); blob.push_back(w.second); }
                P.o_twN = (int)blob.size();
                for (int k = 0; k < Mc; k++) { auto w = W(k, nfft); blob.push_back(w.first); blob.push_back(w.second); }
                P.o_win = (int)blob.size();
                for (int j = 0; j < P.nw; j++) blob.push_back(window[(size_t)j]);
                if (blob.size() & 1) blob.push_back(0.0);
                P.o_winR = (int)blob.size();
                for (int k = 0; k <= P.bix; k++) blob.push_back(windowR[(size_t)k]);
                if (blob.size() & 1) blob.push_back(0.0);
                P.blob_f64 = (int)blob.size();
                PCE_HIP(c, c->pi_blob.reserve(sizeof(double) * blob.size()));
                PCE_HIP(c, hipMemcpyAsync(c->pi_blob.p, blob.data(), sizeof(double) * blob.size(), hipMemcpyHostToDevice, c->stream));
            }
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
            for (int64_t f = 0; f < nf; f += P.fpb) work.push_back({i, (int32_t)f});
        }
        int np2 = 1; while (np2 < max_frames && np2 < PI_MEDIAN_LDS) np2 <<= 1;
        c->pi_np2 = np2;                                       // the in-LDS sort's size; longer slices (a recording of more than 82 s at floor 150) go to k_pitch_median_long
        c->pi_long_slices = max_frames > PI_MEDIAN_LDS;
        c->pi_n_work = (int64_t)work.size();
        PCE_HIP(c, c->pi_meta.reserve(sizeof(PiSlice) * hs.size()));
        PCE_HIP(c, c->pi_work.reserve(sizeof(PiWork) * (work.size() + 1)));
        PCE_HIP(c, c->pi_cand.reserve(sizeof(double) * 32 * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_gpeak.reserve((sizeof(int) + sizeof(double)) * (size_t)(total + 1)));   // ncand + intensity
        PCE_HIP(c, c->pi_psi.reserve((size_t)PI_MAXC * (size_t)(total + 1) + 16));
        PCE_HIP(c, c->pi_f0.reserve(sizeof(double) * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_strength.reserve(sizeof(double) * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_summary.reserve(sizeof(PiSummaryDev) * hs.size()));
        PCE_HIP(c, c->pi_runs.reserve(sizeof(unsigned int) * RUN_LISTS * RF_CSTRIDE + sizeof(PathRun) * (size_t)RUN_LISTS * (size_t)(total / RUN_LISTS + 64)));
        PCE_HIP(c, c->pi_fslice.reserve(sizeof(int) * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_dl.reserve(sizeof(double) * 2 * PI_MAXC * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_rr.reserve(sizeof(double) * (size_t)P.rr_half * (size_t)(total + 1)));
        PCE_HIP(c, c->pi_items.reserve(sizeof(RefineItem) * (size_t)(PI_MAXC - 1) * (size_t)(2 * PI_FPB) * (size_t)(div_up((int64_t)work.size() + 8, RF_LISTS) * RF_LISTS + RF_LISTS)
                                       + sizeof(unsigned int) * RF_LISTS * RF_CSTRIDE));
        PCE_HIP(c, hipMemcpyAsync(c->pi_meta.p, hs.data(), sizeof(PiSlice) * hs.size(), hipMemcpyHostToDevice, c->stream));
        if (!work.empty())
            PCE_HIP(c, hipMemcpyAsync(c->pi_work.p, work.data(), sizeof(PiWork) * work.size(), hipMemcpyHostToDevice, c->stream));
        std::vector<int> fslice((size_t)total + 1, 0);
        for (int32_t i = 0; i < n; i++)
            for (int64_t f = c->pi_frame_off[(size_t)i]; f < c->pi_frame_off[(size_t)i + 1]; f++) fslice[(size_t)f] = i;
        PCE_HIP(c, hipMemcpyAsync(c->pi_fslice.p, fslice.data(), sizeof(int) * fslice.size(), hipMemcpyHostToDevice, c->stream));
        PCE_HIP(c, hipStreamSynchronize(c->stream));
        rc = pce_energy_plan(c, slices, n, c->pi_peakwork, c->pi_acc, &c->pi_n_energy_work);
        if (rc) return rc;
        c->pi_cache.store(slices, n);
        c->pi_params = *p; c->pi_params_valid = true;
    }
    PiParams P; memcpy(&P, c->pi_P, sizeof P);
    hipStream_t tail = c->stream;
    const int64_t total = c->pi_total_frames;
    if (total > 0) {
        // the slice sums and extrema (Praat's mean subtraction and global peak): when pce_energy_run has just produced them
        // for this very slice list they are read from its accumulators (stream ordered) instead of streaming the batch again
        const bool reuse = c->en_n == n && c->en_cache.same(slices, n);
        if (!reuse) {
            int rc = pce_energy_launch(c, n, 500, c->pi_n_energy_work, c->pi_peakwork, c->pi_acc);
            if (rc) return rc;
        }
        size_t stride; const long long *a_sum; const int *a_hi, *a_lo;
        pce_energy_range_ptrs(reuse ? c->en_out : c->pi_acc, &stride, &a_sum, &a_hi, &a_lo);
        double *intensity = c->pi_gpeak.as<double>();
        int *ncand = reinterpret_cast<int *>(intensity + (total + 1));
        {
            int64_t nb = c->pi_n_work;
            nb = (nb + 7) & ~(int64_t)7;         // multiple of 8 so the XCD remap is a bijection; extra blocks exit
            // long windows (44.1 kHz at a 75 Hz floor: 72 KB per wavefront) run fewer wavefronts per workgroup
            int wpb = PI_WPB;
            while (wpb > 1 && sizeof(double) * 4 * (size_t)P.zlen * (size_t)wpb + sizeof(int16_t) * (size_t)P.pcm_span > 160 * 1024) wpb >>= 1;
            if (P.mode != 0) wpb = PI_WPB;
            const size_t lds = P.mode != 0 ? sizeof(double) * ((size_t)reg_wave_f64(P.mode) * PI_WPB + (P.tabs ? (size_t)P.blob_f64 : 0)) + sizeof(int16_t) * (size_t)P.pcm_span
                                           : sizeof(double) * 4 * (size_t)P.zlen * (size_t)wpb + sizeof(int16_t) * (size_t)P.pcm_span;
            const void *kfn = P.mode == 1 ? (P.tabs ? reinterpret_cast<const void *>(k_pitch_frames<4, 1, true>) : reinterpret_cast<const void *>(k_pitch_frames<4, 1, false>))
                              : P.mode == 2 ? (P.tabs ? reinterpret_cast<const void *>(k_pitch_frames<4, 2, true>) : reinterpret_cast<const void *>(k_pitch_frames<4, 2, false>))
                              : P.mode == 3 ? reinterpret_cast<const void *>(k_pitch_frames<4, 3, false>)
                              : wpb == 4 ? reinterpret_cast<const void *>(k_pitch_frames<4, 0, false>)
                              : wpb == 2 ? reinterpret_cast<const void *>(k_pitch_frames<2, 0, false>)
                                         : reinterpret_cast<const void *>(k_pitch_frames<1, 0, false>);
            if (lds > 64 * 1024)
                PCE_HIP(c, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const size_t cnt_bytes = sizeof(unsigned int) * RF_LISTS * RF_CSTRIDE;
            RefineItem *items = reinterpret_cast<RefineItem *>(c->pi_items.as<char>() + cnt_bytes);
            unsigned int *item_count = c->pi_items.as<unsigned int>();
            const unsigned int list_cap = (unsigned int)(div_up(nb, RF_LISTS) * P.fpb * (PI_MAXC - 1));
            PCE_HIP(c, hipMemsetAsync(item_count, 0, cnt_bytes, c->stream));
            {
                // algorithmic fp64 work of a frame (Praat's formulation): mean removal + window (3 nw), two real FFTs of nfft points
                // (2.5 nfft log2 nfft each), the power spectrum (3 nfft / 2), the window-autocorrelation division (maxlag)
                const double f_flops = 3.0 * P.nw + 5.0 * P.nfft * std::log2((double)P.nfft) + 1.5 * P.nfft + (double)P.maxlag;
                KernelTimer t(c, PCE_K_PITCH_FRAMES, nullptr, f_flops * (double)total);
                auto launch = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(64 * wpb), lds, c->stream, c->d_pcm,
                                       c->pi_meta.as<PiSlice>(), c->pi_work.as<PiWork>(), (int)c->pi_n_work, P,
                                       c->pi_window.as<double>(), c->pi_windowR.as<double>(), c->pi_tw.as<double2>(),
                                       c->pi_tw.as<double2>() + (P.nfft >> 1), a_sum, a_hi, a_lo, stride,
                                       c->pi_cand.as<double>(), ncand, intensity, c->pi_rr.as<double>(), items, item_count, list_cap,
                                       c->pi_blob.as<double>());
                };
                if (P.mode == 1) launch(k_pitch_frames<4, 1, false>);
                else if (P.mode == 2) launch(k_pitch_frames<4, 2, true>);
                else if (P.mode == 3) launch(k_pitch_frames<4, 3, false>);
                else if (wpb == 4) launch(k_pitch_frames<4, 0, false>);
                else if (wpb == 2) launch(k_pitch_frames<2, 0, false>);
                else launch(k_pitch_frames<1, 0, false>);
            }
            {
                KernelTimer t(c, PCE_K_PITCH_REFINE);
                // 12 wavefronts per CU are resident (168 VGPRs); 96 per CU measured best (the lists and the iteration counts are uneven), as
                // ONE-wavefront workgroups: a slot is free again when its wavefront ends, not when the slowest of four does (0.98 -> 0.89-0.93 ms).
                // Tried and dropped in round 3: 4 or 16 lanes per candidate (1.66 / 1.02 ms against 0.98); the refinement inside
                // k_pitch_frames, on the autocorrelation still in LDS (no hand-off through HBM: -700 MB per C2 step): 3.07 ms for the
                // fused kernel against 0.95 + 0.98 -- a wavefront's two frames hold ~11 candidates, so its eight lane groups run two rounds of
                // 3-4 dependent evaluations at 70 % occupancy of the lanes, serialised behind its own transforms, where this kernel packs
                // eight candidates of any frames into every wavefront.
                const unsigned blocks = (unsigned)(c->cu_count > 0 ? c->cu_count : 256) * 96u;
                hipLaunchKernelGGL(k_pitch_refine<8>, dim3(blocks), dim3(64), 0, c->stream, P, c->pi_rr.as<double>(), items, item_count,
                                   list_cap, c->pi_cand.as<double>());
            }
        }
        {
            const size_t rc_bytes = sizeof(unsigned int) * RUN_LISTS * RF_CSTRIDE;
            unsigned int *run_count = c->pi_runs.as<unsigned int>();
            PathRun *runs = reinterpret_cast<PathRun *>(c->pi_runs.as<char>() + rc_bytes);
            const unsigned int run_cap = (unsigned int)(total / RUN_LISTS + 64);
            // everything after the refinement (local terms, path finder, median) goes to the side stream: it is HBM- and
            // latency-bound, and whatever the caller launches next on the main stream runs beside it; every consumer
            // joins first (pce_side_join)
            { int rc2 = pce_side_begin(c, pce_ctx::SIDE_TAIL, &tail); if (rc2) return rc2; }
            PCE_HIP(c, hipMemsetAsync(run_count, 0, rc_bytes, tail));
            const long long ne = total * PI_MAXC;
            {
                KernelTimer t(c, PCE_K_PITCH_DELTA, tail);
                hipLaunchKernelGGL(k_pitch_delta, dim3((unsigned)div_up(ne, 256)), dim3(256), 0, tail, P, c->pi_meta.as<PiSlice>(),
                                   c->pi_fslice.as<int>(), c->pi_cand.as<double>(), ncand, intensity, (long long)total,
                                   c->pi_dl.as<double2>(), c->pi_f0.as<double>(), c->pi_strength.as<double>(), runs, run_count, run_cap);
            }
            {
                KernelTimer t(c, PCE_K_PITCH_PATH, tail);
                const unsigned blocks = (unsigned)(c->cu_count > 0 ? c->cu_count : 256) * 16u;     // multiple of RUN_LISTS
                hipLaunchKernelGGL(k_pitch_path, dim3(blocks), dim3(64), 0, tail, P,
                                   c->pi_cand.as<double>(), ncand, c->pi_dl.as<double2>(), runs, run_count, run_cap,
                                   c->pi_psi.as<unsigned char>(), c->pi_f0.as<double>(), c->pi_strength.as<double>());
            }
        }
    }
    if (n > 0) {
        const size_t lds = sizeof(double) * (size_t)c->pi_np2;
        if (lds > 64 * 1024)
            PCE_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pitch_median), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KernelTimer t(c, PCE_K_PITCH_MEDIAN, tail);
        hipLaunchKernelGGL(k_pitch_median, dim3((unsigned)n), dim3(256), lds, tail, c->pi_meta.as<PiSlice>(),
                           c->pi_f0.as<double>(), c->pi_np2, c->pi_summary.as<PiSummaryDev>());
        if (c->pi_long_slices)
            hipLaunchKernelGGL(k_pitch_median_long, dim3((unsigned)n), dim3(256), 0, tail, c->pi_meta.as<PiSlice>(), c->pi_f0.as<double>(), c->pi_np2,
                               c->pi_summary.as<PiSummaryDev>());
    }
    { int rc2 = pce_side_end(c, pce_ctx::SIDE_TAIL, tail); if (rc2) return rc2; }
    PCE_HIP(c, hipGetLastError());
    c->pi_n = n;
    return PCE_OK;
}

int pce_pitch_fetch(pce_ctx *c, double *f0, double *strength, pce_pitch_summary *summary)
{
    if (!c) return PCE_E_INVALID;
    if (c->pi_n < 0) return pce_fail(c, PCE_E_STATE, "pce_pitch_fetch before pce_pitch_run");
    PCE_HIP(c, hipSetDevice(c->device));
    { int rc = pce_side_join(c, pce_ctx::SIDE_TAIL); if (rc) return rc; }
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

size_t pce_pitch_stage_bytes(const pce_ctx *c) { return c->pi_n > 0 ? sizeof(PiSummaryDev) * (size_t)c->pi_n : 0; }
int pce_pitch_stage_enqueue(pce_ctx *c, void *pinned, hipStream_t on)
{
    if (c->pi_n > 0)
        PCE_HIP(c, hipMemcpyAsync(pinned, c->pi_summary.p, sizeof(PiSummaryDev) * (size_t)c->pi_n, hipMemcpyDeviceToHost, on ? on : c->stream));
    return PCE_OK;
}
void pce_pitch_stage_unpack(const void *pinned, int32_t n, pce_pitch_summary *out)
{
    const PiSummaryDev *sd = static_cast<const PiSummaryDev *>(pinned);
    for (int32_t i = 0; i < n; i++) { out[i].n_voiced = sd[i].n_voiced; out[i].median_f0 = sd[i].median; out[i].mean_log_f0 = sd[i].mean_log; }
}

