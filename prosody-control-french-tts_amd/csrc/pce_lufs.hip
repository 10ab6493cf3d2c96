// pce_lufs.hip -- BS.1770 integrated loudness of peak-normalised slices (R4).
//
// Replaces, per slice, what get_lufs (Code/audioPipeline.py:338-358) delegates to
// pyloudnorm: x/peak -> high-shelf biquad -> high-pass biquad (direct form II
// transposed, as scipy.signal.lfilter) -> mean squares of 400 ms blocks with 75 %
// overlap -> absolute (-70 LKFS) and relative (-10 LU) gates -> LUFS.
//
// The IIR recurrence is sequential in time, so a slice is cut into chunks whose
// boundaries contain every gating-block boundary pyloudnorm computes
// (int(T_g*(j*step)*rate), int(T_g*(j*step+1)*rate)) and which are at most LU_LMAX
// samples long.  The filter cascade is linear, so with s the 4-element state
//     state_end(chunk) = A^len * state_begin(chunk) + zero_state_response_end(chunk)
// and three passes make every chunk independent:
//   k_lufs_pass1  one thread per chunk: run the cascade from a zero state, keep the end state
//   k_lufs_scan   one wavefront per slice: propagate true begin states through its chunks (grouped scan)
//   k_lufs_pass2  one thread per chunk: run again from the true state, sum y^2
//   k_lufs_gate   one thread per slice: block energies = sums of whole chunks, gating, LUFS
// All arithmetic is fp64 with contraction off.  Bound: fp64 VALU (about 70 dependent
// flops per sample); algorithmic HBM traffic is 2 B/sample read twice (L2-resident
// the second time).
#include "pce_internal.h"
#include <algorithm>
#include <cmath>

int pce_energy_plan(pce_ctx *c, const pce_slice *slices, int32_t n, DevBuf &work_buf, DevBuf &out_buf, int64_t *n_work);
int pce_energy_launch(pce_ctx *c, int32_t n, int32_t loud_thr, int64_t n_work, DevBuf &work_buf, DevBuf &out_buf, hipStream_t on = nullptr);
const int *pce_energy_peak_ptr(const DevBuf &out_buf, size_t *stride_bytes);

namespace {

constexpr int LU_LMAX = 256;

struct LuChunk { int64_t rel; int32_t len; int32_t slice; };
struct LuSlice {
    int64_t begin, clip_len, clip_off;     // slice start in clip coordinates, clip extent
    int32_t first_chunk, n_chunks;
    int32_t first_block, n_blocks;
    int32_t status, pad;
};
struct LuBlock { int32_t c0, c1; };          // chunk range [c0, c1) of one gating block (indices local to the slice)
struct LuCoef { double b[3], a[3], c[3], d[3]; double inv_norm; };   // stage 1 (b,a), stage 2 (c,d), 1/(T_g*rate)

// One chunk of the cascade.  Samples are fetched 8 at a time with aligned 16-byte loads (a lane's
// chunk is a contiguous run of int16; per-sample 2-byte loads cost 2.7x the HBM traffic in PMC
// counters); samples outside the clip are the slice's virtual zeros.
template <bool SUM>
__device__ __forceinline__ double lu_run(const int16_t *__restrict__ pcm, const LuSlice &s, const LuChunk &ch, const LuCoef &k,
                                         double peak, double st[4], int64_t pcm_total)
{
    double z0 = st[0], z1 = st[1], w0 = st[2], w1 = st[3], acc = 0.0;
    const int64_t lo = s.clip_off, hi = s.clip_off + s.clip_len;          // real samples: global index in [lo, hi)
    const int64_t g0 = s.clip_off + s.begin + ch.rel, g1 = g0 + ch.len;   // this chunk, global indices
    for (int64_t a = g0 & ~(int64_t)7; a < g1; a += 8) {
        int4 v = make_int4(0, 0, 0, 0);
        if (a >= 0 && a < pcm_total) v = *reinterpret_cast<const int4 *>(pcm + a);
        const int words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int64_t g = a + q;
            if (g < g0 || g >= g1) continue;
            const int raw = (q & 1) ? (words[q >> 1] >> 16) : (int)(short)(words[q >> 1] & 0xFFFF);
            const double x = ((g >= lo && g < hi) ? (double)raw : 0.0) / peak;
            const double y = k.b[0] * x + z0;
            z0 = k.b[1] * x - k.a[1] * y + z1;
            z1 = k.b[2] * x - k.a[2] * y;
            const double u = k.c[0] * y + w0;
            w0 = k.c[1] * y - k.d[1] * u + w1;
            w1 = k.c[2] * y - k.d[2] * u;
            if (SUM) acc += u * u;
        }
    }
    st[0] = z0; st[1] = z1; st[2] = w0; st[3] = w1;
    return acc;
}

__device__ __forceinline__ double lu_peak(const int *peaks, size_t stride, int slice)
{
    const int p = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(peaks) + stride * (size_t)slice);
    return p == 0 ? 1.0 : (double)p;       // `np.abs(samples).max() or 1.0`
}

__global__ void k_lufs_pass1(const int16_t *__restrict__ pcm, const LuSlice *__restrict__ slices, const LuChunk *__restrict__ chunks,
                             int n_chunks, LuCoef k, const int *peaks, size_t pstride, double *__restrict__ state_end, int64_t pcm_total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks) return;
    const LuChunk ch = chunks[i];
    const LuSlice s = slices[ch.slice];
    double st[4] = {0.0, 0.0, 0.0, 0.0};
    lu_run<false>(pcm, s, ch, k, lu_peak(peaks, pstride, ch.slice), st, pcm_total);
    double *o = state_end + 4 * (size_t)i;
    o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3];
}

__device__ __forceinline__ double lu_readlane_f64(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// One wavefront per slice, three levels (the plain chain over a 10 s slice is 625 dependent 4x4
// matrix-vector steps; this is 16 + n_groups + 16):
//   1. lane g folds its group of LU_GROUP consecutive chunks from a zero state: z_g, and the
//      group's transition matrix M_g = prod A^len;
//   2. the group begin-states s_(g+1) = M_g s_g + z_g are chained wave-uniform (v_readlane);
//   3. lane g replays its group from s_g and writes every chunk's begin state.
constexpr int LU_GROUP = 16;
__device__ __forceinline__ void lu_matvec(const double *__restrict__ m, double &s0, double &s1, double &s2, double &s3,
                                          double e0, double e1, double e2, double e3)
{
    const double n0 = (fma(m[1], s1, fma(m[0], s0, e0))) + (fma(m[3], s3, m[2] * s2));
    const double n1 = (fma(m[5], s1, fma(m[4], s0, e1))) + (fma(m[7], s3, m[6] * s2));
    const double n2 = (fma(m[9], s1, fma(m[8], s0, e2))) + (fma(m[11], s3, m[10] * s2));
    const double n3 = (fma(m[13], s1, fma(m[12], s0, e3))) + (fma(m[15], s3, m[14] * s2));
    s0 = n0; s1 = n1; s2 = n2; s3 = n3;
}
__global__ __launch_bounds__(64) void k_lufs_scan(const LuSlice *__restrict__ slices, const LuChunk *__restrict__ chunks, int n_slices,
                                                  const double *__restrict__ apow /* [LU_LMAX+1][16] */,
                                                  const double *__restrict__ state_end, double *__restrict__ state_init)
{
    const int i = blockIdx.x;
    if (i >= n_slices) return;
    const int lane = threadIdx.x;
    const LuSlice s = slices[i];
    double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;                      // state at the start of the current 64-group block
    for (int g0 = 0; g0 * LU_GROUP < s.n_chunks; g0 += 64) {
        const int g = g0 + lane;
        const int cb = g * LU_GROUP, ce = min(cb + LU_GROUP, s.n_chunks);   // this lane's chunks [cb, ce)
        // level 1: zero-state response and transition matrix of the group
        double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
        double M[16];
#pragma unroll
        for (int t = 0; t < 16; t++) M[t] = (t % 5 == 0) ? 1.0 : 0.0;
        for (int c = cb; c < ce; c++) {
            const size_t ci = (size_t)(s.first_chunk + c);
            const double *e = state_end + 4 * ci;
            const double *a = apow + (size_t)chunks[ci].len * 16;
            double A[16];
#pragma unroll
            for (int t = 0; t < 16; t++) A[t] = a[t];
            lu_matvec(A, z0, z1, z2, z3, e[0], e[1], e[2], e[3]);
            double N[16];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    N[r * 4 + q] = fma(A[r * 4 + 1], M[4 + q], A[r * 4] * M[q]) + fma(A[r * 4 + 3], M[12 + q], A[r * 4 + 2] * M[8 + q]);
#pragma unroll
            for (int t = 0; t < 16; t++) M[t] = N[t];
        }
        // level 2: begin state of every group of this block (wave-uniform chain over the lanes)
        const int ng = min(64, (s.n_chunks - g0 * LU_GROUP + LU_GROUP - 1) / LU_GROUP);
        double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        for (int q = 0; q < ng; q++) {
            if (lane == q) { b0 = c0; b1 = c1; b2 = c2; b3 = c3; }
            double Mq[16];
#pragma unroll
            for (int t = 0; t < 16; t++) Mq[t] = lu_readlane_f64(M[t], q);
            lu_matvec(Mq, c0, c1, c2, c3, lu_readlane_f64(z0, q), lu_readlane_f64(z1, q), lu_readlane_f64(z2, q), lu_readlane_f64(z3, q));
        }
        // level 3: replay the group from its true begin state
        for (int c = cb; c < ce; c++) {
            const size_t ci = (size_t)(s.first_chunk + c);
            double *o = state_init + 4 * ci;
            o[0] = b0; o[1] = b1; o[2] = b2; o[3] = b3;
            const double *e = state_end + 4 * ci;
            const double *a = apow + (size_t)chunks[ci].len * 16;
            double A[16];
#pragma unroll
            for (int t = 0; t < 16; t++) A[t] = a[t];
            lu_matvec(A, b0, b1, b2, b3, e[0], e[1], e[2], e[3]);
        }
    }
}

__global__ void k_lufs_pass2(const int16_t *__restrict__ pcm, const LuSlice *__restrict__ slices, const LuChunk *__restrict__ chunks,
                             int n_chunks, LuCoef k, const int *peaks, size_t pstride, const double *__restrict__ state_init,
                             double *__restrict__ energy, int64_t pcm_total)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks) return;
    const LuChunk ch = chunks[i];
    const LuSlice s = slices[ch.slice];
    const double *in = state_init + 4 * (size_t)i;
    double st[4] = {in[0], in[1], in[2], in[3]};
    energy[i] = lu_run<true>(pcm, s, ch, k, lu_peak(peaks, pstride, ch.slice), st, pcm_total);
}

__device__ __forceinline__ double lu_wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One wavefront per slice: lanes own gating blocks (block energies = sums of whole chunks,
// in chunk order), the two gated means are butterfly reductions.
__global__ __launch_bounds__(64) void k_lufs_gate(const LuSlice *__restrict__ slices, const LuBlock *__restrict__ blocks, int n_slices, LuCoef k,
                                                  const double *__restrict__ energy, double *__restrict__ zbuf, double *__restrict__ lufs)
{
    const int i = blockIdx.x;
    if (i >= n_slices) return;
    const int lane = threadIdx.x;
    const LuSlice s = slices[i];
    if (s.status != PCE_SLICE_OK) { if (lane == 0) lufs[i] = nan(""); return; }
    const double *en = energy + s.first_chunk;
    double *z = zbuf + s.first_block;
    const double gamma_a = -70.0;
    double sum = 0.0, cnt = 0.0;
    for (int j = lane; j < s.n_blocks; j += 64) {
        const LuBlock b = blocks[s.first_block + j];
        double e = 0.0;
        for (int c = b.c0; c < b.c1; c++) e += en[c];
        const double zj = k.inv_norm * e;
        z[j] = zj;
        const double lj = -0.691 + 10.0 * log10(zj);
        if (lj >= gamma_a) { sum += zj; cnt += 1.0; }
    }
    sum = lu_wave_sum(sum); cnt = lu_wave_sum(cnt);
    const double zavg1 = cnt > 0.0 ? sum / cnt : nan("");
    const double gamma_r = -0.691 + 10.0 * log10(zavg1) - 10.0;
    sum = 0.0; cnt = 0.0;
    for (int j = lane; j < s.n_blocks; j += 64) {
        const double zj = z[j];
        const double lj = -0.691 + 10.0 * log10(zj);
        if (lj > gamma_r && lj > gamma_a) { sum += zj; cnt += 1.0; }
    }
    sum = lu_wave_sum(sum); cnt = lu_wave_sum(cnt);
    const double zavg2 = cnt > 0.0 ? sum / cnt : 0.0;
    if (lane == 0) lufs[i] = -0.691 + 10.0 * log10(zavg2);
}

// pyloudnorm.IIRfilter.generate_coefficients for the two K-weighting stages
void kweight_design(double rate, LuCoef &k)
{
    const double PI = 3.14159265358979323846;
    {
        const double G = 4.0, Q = 1.0 / std::sqrt(2.0), fc = 1500.0;
        const double A = std::pow(10.0, G / 40.0), w0 = 2.0 * PI * (fc / rate), alpha = std::sin(w0) / (2.0 * Q);
        const double b0 = A * ((A + 1) + (A - 1) * std::cos(w0) + 2 * std::sqrt(A) * alpha);
        const double b1 = -2 * A * ((A - 1) + (A + 1) * std::cos(w0));
        const double b2 = A * ((A + 1) + (A - 1) * std::cos(w0) - 2 * std::sqrt(A) * alpha);
        const double a0 = (A + 1) - (A - 1) * std::cos(w0) + 2 * std::sqrt(A) * alpha;
        const double a1 = 2 * ((A - 1) - (A + 1) * std::cos(w0));
        const double a2 = (A + 1) - (A - 1) * std::cos(w0) - 2 * std::sqrt(A) * alpha;
        k.b[0] = b0 / a0; k.b[1] = b1 / a0; k.b[2] = b2 / a0;
        k.a[0] = a0 / a0; k.a[1] = a1 / a0; k.a[2] = a2 / a0;
    }
    {
        const double Q = 0.5, fc = 38.0;
        const double w0 = 2.0 * PI * (fc / rate), alpha = std::sin(w0) / (2.0 * Q);
        const double b0 = (1 + std::cos(w0)) / 2, b1 = -(1 + std::cos(w0)), b2 = (1 + std::cos(w0)) / 2;
        const double a0 = 1 + alpha, a1 = -2 * std::cos(w0), a2 = 1 - alpha;
        k.c[0] = b0 / a0; k.c[1] = b1 / a0; k.c[2] = b2 / a0;
        k.d[0] = a0 / a0; k.d[1] = a1 / a0; k.d[2] = a2 / a0;
    }
    k.inv_norm = 1.0 / (0.400 * rate);
}

// state-transition matrix of the cascade with zero input, and its powers 0..LU_LMAX
void transition_powers(const LuCoef &k, std::vector<double> &pw)
{
    const double A[16] = {
        -k.a[1], 1.0, 0.0, 0.0,
        -k.a[2], 0.0, 0.0, 0.0,
        k.c[1] - k.d[1] * k.c[0], 0.0, -k.d[1], 1.0,
        k.c[2] - k.d[2] * k.c[0], 0.0, -k.d[2], 0.0};
    pw.assign((size_t)(LU_LMAX + 1) * 16, 0.0);
    for (int i = 0; i < 4; i++) pw[(size_t)i * 4 + i] = 1.0;
    for (int L = 1; L <= LU_LMAX; L++) {
        const double *prev = &pw[(size_t)(L - 1) * 16];
        double *cur = &pw[(size_t)L * 16];
        for (int r = 0; r < 4; r++)
            for (int q = 0; q < 4; q++) {
                double s = 0.0;
                for (int t = 0; t < 4; t++) s += A[r * 4 + t] * prev[t * 4 + q];
                cur[r * 4 + q] = s;
            }
    }
}

} // namespace

static int lufs_plan(pce_ctx *c, const pce_slice *slices, int32_t n)
{
    std::vector<LuSlice> hs((size_t)(n > 0 ? n : 1));
    std::vector<LuChunk> chunks;
    std::vector<LuBlock> blocks;
    const double rate = (double)(c->lu_meter_rate > 0 ? c->lu_meter_rate : c->rate);   // pyloudnorm.Meter(rate): the METER's rate, not necessarily the data's
    const double T_g = 0.400, step = 1.0 - 0.75;
    c->lu_host_status.assign((size_t)n, PCE_SLICE_OK);
    std::vector<int64_t> pts;
    for (int32_t i = 0; i < n; i++) {
        const pce_slice &s = slices[i];
        if (s.clip < 0 || s.clip >= c->n_clips) return pce_fail(c, PCE_E_INVALID, "slice %d: clip %d out of range", i, s.clip);
        if (s.end < s.begin) return pce_fail(c, PCE_E_INVALID, "slice %d: end < begin", i);
        LuSlice &h = hs[(size_t)i];
        h.begin = s.begin; h.clip_off = c->clip_off[s.clip]; h.clip_len = c->clip_off[s.clip + 1] - c->clip_off[s.clip];
        h.first_chunk = (int32_t)chunks.size(); h.n_chunks = 0;
        h.first_block = (int32_t)blocks.size(); h.n_blocks = 0;
        h.status = PCE_SLICE_OK; h.pad = 0;
        const int64_t ns = s.end - s.begin;
        if (ns == 0) { h.status = PCE_SLICE_EMPTY; c->lu_host_status[(size_t)i] = h.status; continue; }
        if ((double)ns < T_g * rate) { h.status = PCE_SLICE_TOO_SHORT; c->lu_host_status[(size_t)i] = h.status; continue; }
        // pyloudnorm: T = numSamples / rate; numBlocks = int(np.round((T - T_g) / (T_g * step)) + 1)
        const double T = (double)ns / rate;
        const int64_t nb = (int64_t)std::nearbyint((T - T_g) / (T_g * step)) + 1;
        pts.clear(); pts.push_back(0);
        std::vector<int64_t> lo((size_t)(nb > 0 ? nb : 0)), up((size_t)(nb > 0 ? nb : 0));
        for (int64_t j = 0; j < nb; j++) {
            int64_t l = (int64_t)(T_g * ((double)j * step) * rate);
            int64_t u = (int64_t)(T_g * ((double)j * step + 1.0) * rate);
            if (l > ns) l = ns;
            if (u > ns) u = ns;
            if (u < l) u = l;
            lo[(size_t)j] = l; up[(size_t)j] = u;
            pts.push_back(l); pts.push_back(u);
        }
        std::sort(pts.begin(), pts.end());
        pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
        // chunks between consecutive points, each at most LU_LMAX long; remember the chunk index of every point
        std::vector<int32_t> chunk_at(pts.size(), 0);
        for (size_t p = 0; p + 1 < pts.size(); p++) {
            chunk_at[p] = (int32_t)chunks.size() - h.first_chunk;
            for (int64_t a = pts[p]; a < pts[p + 1]; a += LU_LMAX) {
                const int64_t len = std::min<int64_t>(LU_LMAX, pts[p + 1] - a);
                chunks.push_back({a, (int32_t)len, i});
            }
        }
        chunk_at[pts.size() - 1] = (int32_t)chunks.size() - h.first_chunk;
        h.n_chunks = (int32_t)chunks.size() - h.first_chunk;
        for (int64_t j = 0; j < nb; j++) {
            const size_t pl = (size_t)(std::lower_bound(pts.begin(), pts.end(), lo[(size_t)j]) - pts.begin());
            const size_t pu = (size_t)(std::lower_bound(pts.begin(), pts.end(), up[(size_t)j]) - pts.begin());
            blocks.push_back({chunk_at[pl], chunk_at[pu]});
        }
        h.n_blocks = (int32_t)nb;
    }
    if (chunks.size() > (size_t)INT32_MAX) return pce_fail(c, PCE_E_LIMIT, "too many LUFS chunks");
    c->lu_n_chunks = (int64_t)chunks.size();
    c->lu_n_blocks = (int64_t)blocks.size();
    LuCoef k; kweight_design(rate, k);
    std::vector<double> pw; transition_powers(k, pw);
    static_assert(sizeof(LuCoef) == sizeof(double) * 13, "LuCoef layout");
    memcpy(c->lu_coef, &k, sizeof k);

    PCE_HIP(c, c->lu_meta.reserve(sizeof(LuSlice) * hs.size()));
    PCE_HIP(c, c->lu_chunks.reserve(sizeof(LuChunk) * (chunks.size() + 1)));
    PCE_HIP(c, c->lu_blocks.reserve(sizeof(LuBlock) * (blocks.size() + 1)));
    PCE_HIP(c, c->lu_pow.reserve(sizeof(double) * pw.size()));
    PCE_HIP(c, c->lu_state_end.reserve(sizeof(double) * 4 * (chunks.size() + 1)));
    PCE_HIP(c, c->lu_state_init.reserve(sizeof(double) * 4 * (chunks.size() + 1)));
    PCE_HIP(c, c->lu_energy.reserve(sizeof(double) * (chunks.size() + 1)));
    PCE_HIP(c, c->lu_zbuf.reserve(sizeof(double) * (blocks.size() + 1)));   // per-block z scratch
    PCE_HIP(c, c->lu_out.reserve(sizeof(double) * hs.size()));
    PCE_HIP(c, hipMemcpyAsync(c->lu_meta.p, hs.data(), sizeof(LuSlice) * hs.size(), hipMemcpyHostToDevice, c->stream));
    if (!chunks.empty())
        PCE_HIP(c, hipMemcpyAsync(c->lu_chunks.p, chunks.data(), sizeof(LuChunk) * chunks.size(), hipMemcpyHostToDevice, c->stream));
    if (!blocks.empty())
        PCE_HIP(c, hipMemcpyAsync(c->lu_blocks.p, blocks.data(), sizeof(LuBlock) * blocks.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(c->lu_pow.p, pw.data(), sizeof(double) * pw.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    // peaks of the same slices
    return pce_energy_plan(c, slices, n, c->lu_en_work, c->lu_en_acc, &c->lu_n_energy_work);
}

extern "C" {

int pce_lufs_set_meter_rate(pce_ctx *c, int32_t rate)
{
    if (!c || rate < 0) return PCE_E_INVALID;
    if (c->lu_meter_rate != rate) { c->lu_meter_rate = rate; c->lu_cache.drop(); }
    return PCE_OK;
}

int pce_lufs_run(pce_ctx *c, const pce_slice *slices, int32_t n)
{
    if (!c || (!slices && n > 0) || n < 0) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    PCE_HIP(c, hipSetDevice(c->device));
    if (!c->lu_cache.same(slices, n)) {
        { int rc = pce_side_join(c, pce_ctx::SIDE_LUFS); if (rc) return rc; }      // the previous chain still uses the plan's buffers
        c->lu_n = -1;
        int st = lufs_plan(c, slices, n);
        if (st) return st;
        c->lu_cache.store(slices, n);
    }
    // The whole chain goes to the second side stream (forked behind whatever the main stream holds: the upload, the
    // previous batch's result copies), so the pitch kernels launched next run beside it; consumers join first.
    hipStream_t ls = c->stream;
    { int rc = pce_side_begin(c, pce_ctx::SIDE_LUFS, &ls); if (rc) return rc; }
    // the peaks that normalise the slices (get_lufs divides by max |x|): when pce_energy_run has just covered the SAME slices its
    // accumulators hold them (the fork above orders this chain behind it): no second pass over the PCM.  pce_energy_run joins this side
    // stream before it overwrites them (lu_reads_en_out).
    const bool reuse = c->en_n == n && c->en_cache.same(slices, n);
    if (!reuse) {
        int st = pce_energy_launch(c, n, 500, c->lu_n_energy_work, c->lu_en_work, c->lu_en_acc, ls);
        if (st) return st;
    }
    c->lu_reads_en_out = reuse;
    LuCoef k; memcpy(&k, c->lu_coef, sizeof k);
    size_t pstride = 0;
    const int *peaks = pce_energy_peak_ptr(reuse ? c->en_out : c->lu_en_acc, &pstride);
    const int nch = (int)c->lu_n_chunks;
    if (nch > 0) {
        {
            KernelTimer t(c, PCE_K_LUFS_PASS1, ls);
            hipLaunchKernelGGL(k_lufs_pass1, dim3((unsigned)div_up(nch, 64)), dim3(64), 0, ls, c->d_pcm,
                               c->lu_meta.as<LuSlice>(), c->lu_chunks.as<LuChunk>(), nch, k, peaks, pstride,
                               c->lu_state_end.as<double>(), (int64_t)c->clip_off[(size_t)c->n_clips]);
        }
        {
            KernelTimer t(c, PCE_K_LUFS_SCAN, ls);
            hipLaunchKernelGGL(k_lufs_scan, dim3((unsigned)n), dim3(64), 0, ls,
                               c->lu_meta.as<LuSlice>(), c->lu_chunks.as<LuChunk>(), (int)n, c->lu_pow.as<double>(),
                               c->lu_state_end.as<double>(), c->lu_state_init.as<double>());
        }
        {
            KernelTimer t(c, PCE_K_LUFS_PASS2, ls);
            hipLaunchKernelGGL(k_lufs_pass2, dim3((unsigned)div_up(nch, 64)), dim3(64), 0, ls, c->d_pcm,
                               c->lu_meta.as<LuSlice>(), c->lu_chunks.as<LuChunk>(), nch, k, peaks, pstride,
                               c->lu_state_init.as<double>(), c->lu_energy.as<double>(), (int64_t)c->clip_off[(size_t)c->n_clips]);
        }
    }
    if (n > 0) {
        KernelTimer t(c, PCE_K_LUFS_GATE, ls);
        hipLaunchKernelGGL(k_lufs_gate, dim3((unsigned)n), dim3(64), 0, ls,
                           c->lu_meta.as<LuSlice>(), c->lu_blocks.as<LuBlock>(), (int)n, k, c->lu_energy.as<double>(),
                           c->lu_zbuf.as<double>(), c->lu_out.as<double>());
    }
    PCE_HIP(c, hipGetLastError());
    { int rc = pce_side_end(c, pce_ctx::SIDE_LUFS, ls); if (rc) return rc; }
    c->lu_n = n;
    return PCE_OK;
}

int pce_lufs_fetch(pce_ctx *c, double *lufs, int32_t *status)
{
    if (!c || !lufs) return PCE_E_INVALID;
    if (c->lu_n < 0) return pce_fail(c, PCE_E_STATE, "pce_lufs_fetch before pce_lufs_run");
    PCE_HIP(c, hipSetDevice(c->device));
    { int rc = pce_side_join(c, pce_ctx::SIDE_LUFS); if (rc) return rc; }
    if (c->lu_n > 0)
        PCE_HIP(c, hipMemcpyAsync(lufs, c->lu_out.p, sizeof(double) * (size_t)c->lu_n, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    if (status) for (int32_t i = 0; i < c->lu_n; i++) status[i] = c->lu_host_status[(size_t)i];
    return PCE_OK;
}

} // extern "C"
