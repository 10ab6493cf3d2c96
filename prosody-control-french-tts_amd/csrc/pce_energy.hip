// pce_energy.hip -- k_energy: exact integer short-time energy / peak / gate counts per slice.
//
// Reference arithmetic replaced (all reference-owned, pinned by goldens G3/G4):
//   Code/Pipeline/compute_loudness_adjustments.py:17-21  samples**2 on int16 (wraps), mean
//   Code/Aligners/use_whisper_timestamped.py:204-210      mean(square(f32)), count(|x| > 500)
//   Code/audioPipeline.py:349                             np.abs(samples).max()
//
// Roofline: HBM-bound, 2 algorithmic bytes per sample read once, O(1) bytes written.
// Layout: int16 PCM, clips concatenated; each 256-thread block streams one chunk of up to
// CHUNK samples with 16-byte loads (8 samples / lane / load, 4 KiB per wave-instruction
// group), reduces in registers -> wave shuffles -> LDS -> one set of integer atomics per
// block.  Integer accumulation makes the result independent of the reduction order.
#include "pce_internal.h"

namespace {

constexpr int EN_THREADS = 256;
// loads per lane in flight = chunk size / 4 KiB: EN_LOADS = 8 (32 KiB chunks; 16 measured no better)
constexpr int EN_LOADS = 8;
static inline int64_t en_chunk(int iters) { return (int64_t)EN_THREADS * 8 * iters; }

struct EnWork { int64_t g0, g1; int32_t slice; int32_t pad; };
// m_hi = max(x + 32769) and m_lo = max(32768 - x) over the real samples (0 = no sample seen):
// zero-initialisable encodings of the slice maximum and minimum, used by the pitch path.
struct EnAcc { unsigned long long sum_sq; long long sum_wrap; unsigned long long n_loud; long long sum; int peak; int m_hi; int m_lo; int pad; };

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_down(v, off, 64));
    return v;
}

// wave-wide integer reductions without LDS round trips: four DPP steps leave every lane of a 16-lane row with the row's total, four
// v_readlane combine the rows on the scalar unit (the result is wave-uniform).  The shuffle form (ds_bpermute, six dependent steps per
// value, seven values, two of them 64-bit) plus the LDS hand-off between waves was the larger part of a workgroup's life after its loads.
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += dpp_i32<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_i32<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_i32<0x141>(v);     // row_half_mirror
    v += dpp_i32<0x140>(v);     // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_max_dpp(int v)
{
    v = max(v, dpp_i32<0xB1>(v));
    v = max(v, dpp_i32<0x4E>(v));
    v = max(v, dpp_i32<0x141>(v));
    v = max(v, dpp_i32<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// A workgroup streams `cpb` CONSECUTIVE chunks (mostly of one slice), one after the other, with its accumulators in registers, and
// adds its share to a slice's accumulators ONCE per (workgroup, slice): wave totals by DPP, one LDS hand-off, seven atomics from one
// lane.  What the measurements of round 3 said about the one-chunk form (400 MB, 93 us): without its atomics 79 us, without the
// per-sample route as well 65 us = 6.2 TB/s, the rate of a bare read loop of this shape with the same arithmetic
// (tools/lab/read_probe.hip: 6.1 TB/s; 6.9 with no arithmetic).  Hence: fewer atomics per byte (cpb chunks per set), loads outside
// the slice skipped before the route is chosen (the last chunk of a clip sent whole waves down the per-sample route for samples that
// are not there), non-temporal loads for a batch beyond the Infinity Cache.
template <int EN_ITERS, bool NT>
__global__ __launch_bounds__(EN_THREADS) void k_energy(const int16_t *__restrict__ pcm, const EnWork *__restrict__ work, int n_work, int cpb,
                                                      int loud_thr, EnAcc *__restrict__ out)
{
    typedef int i4 __attribute__((ext_vector_type(4)));
    typedef short s2 __attribute__((ext_vector_type(2)));
    __shared__ int l_part[EN_THREADS / 64][8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i0 = blockIdx.x * cpb, i1 = min(i0 + cpb, n_work);
    unsigned long long s_sq = 0; int s_wrap = 0; int n_loud = 0; int peak = 0; int s_sum = 0; int m_hi = 0, m_lo = 0;
    const s2 ones = {1, 1};
    s2 pmax = {-32768, -32768}, pmin = {32767, 32767};
    bool any_packed = false;
    for (int i = i0; i < i1; i++) {
        const EnWork w = work[i];
        // all EN_ITERS 16-byte loads of this lane are issued before the first is consumed (a chunk is at most
        // EN_ITERS * 4 KiB per wave): the kernel is a pure stream, latency hides only behind bytes in flight
        i4 v[EN_ITERS];
        const int64_t p0 = (w.g0 & ~(int64_t)7) + (int64_t)threadIdx.x * 8;
#pragma unroll
        for (int it = 0; it < EN_ITERS; it++) {
            const int64_t pos = p0 + (int64_t)it * EN_THREADS * 8;
            v[it] = (i4){0, 0, 0, 0};
            if (pos < w.g1) v[it] = NT ? __builtin_nontemporal_load(reinterpret_cast<const i4 *>(pcm + pos)) : *reinterpret_cast<const i4 *>(pcm + pos);
        }
        // Loads that lie wholly inside the slice (all but the first and last of a chunk) take the packed route: two samples
        // per instruction on v_dot2_i32_i16 (x0^2 + x1^2, x0 + x1, the sum of the two wrapped squares), v_pk_mul_lo_u16 (the
        // int16-wrapped squares themselves) and v_pk_max_i16 / v_pk_min_i16 (extrema); the per-sample route handles edges.
#pragma unroll
        for (int it = 0; it < EN_ITERS; it++) {
            const int64_t pos = p0 + (int64_t)it * EN_THREADS * 8;
            if (pos + 8 <= w.g0 || pos >= w.g1) continue;        // nothing of this load belongs to the slice
            const int words[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
            if (pos >= w.g0 && pos + 8 <= w.g1) {
                any_packed = true;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const s2 xv = __builtin_bit_cast(s2, words[k]);
                    s_sq += (unsigned long long)(unsigned int)__builtin_amdgcn_sdot2(xv, xv, 0, false);      // <= 2^31: exact as unsigned
                    s_sum = __builtin_amdgcn_sdot2(xv, ones, s_sum, false);
                    s_wrap = __builtin_amdgcn_sdot2(xv * xv, ones, s_wrap, false);                           // (int16)(x^2) per half, summed
                    pmax = __builtin_elementwise_max(pmax, xv); pmin = __builtin_elementwise_min(pmin, xv);
                    const s2 ab = __builtin_elementwise_max(xv, (s2){0, 0} - xv);                             // |x| as int16: |-32768| wraps to -32768
                    n_loud += ((int)ab.x > loud_thr) + ((int)ab.y > loud_thr);
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int lo = (int)(short)(words[k] & 0xFFFF), hi = words[k] >> 16;
                const int64_t j0 = pos + 2 * k, j1 = j0 + 1;
                const bool ok0 = j0 >= w.g0 && j0 < w.g1, ok1 = j1 >= w.g0 && j1 < w.g1;
                const int x0 = ok0 ? lo : 0, x1 = ok1 ? hi : 0;
                const unsigned int q0 = (unsigned int)(x0 * x0), q1 = (unsigned int)(x1 * x1);
                s_sq += (unsigned long long)q0 + (unsigned long long)q1;
                s_wrap += (int)(short)(q0 & 0xFFFFu) + (int)(short)(q1 & 0xFFFFu);
                const int ab0 = x0 < 0 ? -x0 : x0, ab1 = x1 < 0 ? -x1 : x1;
                peak = max(peak, max(ab0, ab1));
                n_loud += ((int)(short)ab0 > loud_thr) + ((int)(short)ab1 > loud_thr);
                s_sum += x0 + x1;
                m_hi = max(m_hi, max(ok0 ? x0 + 32769 : 0, ok1 ? x1 + 32769 : 0));
                m_lo = max(m_lo, max(ok0 ? 32768 - x0 : 0, ok1 ? 32768 - x1 : 0));
            }
        }
        if (i + 1 < i1 && work[i + 1].slice == w.slice) continue;              // (workgroup-uniform) the next chunk adds to the same slice
        if (any_packed) {
            const int hi = max((int)pmax.x, (int)pmax.y), lo = min((int)pmin.x, (int)pmin.y);
            peak = max(peak, max(hi < 0 ? -hi : hi, lo < 0 ? -lo : lo));
            m_hi = max(m_hi, hi + 32769); m_lo = max(m_lo, 32768 - lo);
        }
        // integer sums and maxima: the result does not depend on the order.  Every WAVE total fits 32 bits: a lane holds at most
        // cpb * EN_ITERS * 8 <= 2^10 samples; sum and wrapped-square sum: 2^10 x 2^15 per lane, x 64 lanes <= 2^31 in magnitude (only
        // -2^31 is reached); the 64-bit sum of squares (< 2^40 per lane) travels as its low 20 bits and the rest: 64 x 2^20 each.
        // The four waves are then added as 64-bit values.
        const int r_sq_lo = wave_sum_dpp((int)(s_sq & 0xFFFFFull)), r_sq_hi = wave_sum_dpp((int)(s_sq >> 20));
        const int r_wrap = wave_sum_dpp(s_wrap), r_loud = wave_sum_dpp(n_loud), r_sum = wave_sum_dpp(s_sum);
        const int r_peak = wave_max_dpp(peak), r_hi = wave_max_dpp(m_hi), r_lo = wave_max_dpp(m_lo);
        if (lane == 0) {
            l_part[wv][0] = r_sq_lo; l_part[wv][1] = r_sq_hi; l_part[wv][2] = r_wrap; l_part[wv][3] = r_loud;
            l_part[wv][4] = r_sum; l_part[wv][5] = r_peak; l_part[wv][6] = r_hi; l_part[wv][7] = r_lo;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t_sq = 0, t_loud = 0; long long t_wrap = 0, t_sum = 0; int t_peak = 0, t_hi = 0, t_lo = 0;
            for (int q = 0; q < EN_THREADS / 64; q++) {
                t_sq += ((unsigned long long)(unsigned int)l_part[q][1] << 20) + (unsigned long long)(unsigned int)l_part[q][0];
                t_wrap += (long long)l_part[q][2]; t_loud += (unsigned long long)(unsigned int)l_part[q][3]; t_sum += (long long)l_part[q][4];
                t_peak = max(t_peak, l_part[q][5]); t_hi = max(t_hi, l_part[q][6]); t_lo = max(t_lo, l_part[q][7]);
            }
            EnAcc *o = out + w.slice;
            atomicAdd(&o->sum_sq, t_sq);
            atomicAdd(reinterpret_cast<unsigned long long *>(&o->sum_wrap), (unsigned long long)t_wrap);
            atomicAdd(&o->n_loud, t_loud);
            atomicMax(&o->peak, t_peak);
            atomicAdd(reinterpret_cast<unsigned long long *>(&o->sum), (unsigned long long)t_sum);
            atomicMax(&o->m_hi, t_hi);
            atomicMax(&o->m_lo, t_lo);
        }
        __syncthreads();                                           // the LDS words are free for the next slice of this workgroup
        s_sq = 0; s_wrap = 0; n_loud = 0; peak = 0; s_sum = 0; m_hi = 0; m_lo = 0;
        pmax = (s2){-32768, -32768}; pmin = (s2){32767, 32767}; any_packed = false;
    }
}

} // namespace

// Plan (host): clamp every slice to its clip and cut it into 16-byte-aligned chunks.
// Shared with the LUFS path, which needs per-slice peaks resident on the device.
int pce_energy_plan(pce_ctx *c, const pce_slice *slices, int32_t n, DevBuf &work_buf, DevBuf &out_buf, int64_t *n_work)
{
    std::vector<EnWork> work;
    for (int32_t i = 0; i < n; i++) {
        const pce_slice &s = slices[i];
        if (s.clip < 0 || s.clip >= c->n_clips) return pce_fail(c, PCE_E_INVALID, "slice %d: clip %d out of range", i, s.clip);
        if (s.end < s.begin) return pce_fail(c, PCE_E_INVALID, "slice %d: end < begin", i);
        const int64_t len = c->clip_off[s.clip + 1] - c->clip_off[s.clip];
        int64_t b = s.begin < 0 ? 0 : s.begin, e = s.end > len ? len : s.end;
        if (e <= b) continue;
        const int64_t g0 = c->clip_off[s.clip] + b, g1 = c->clip_off[s.clip] + e;
        const int64_t EN_CHUNK = en_chunk(EN_LOADS);
        for (int64_t p = g0; p < g1;) {
            int64_t q = ((p / EN_CHUNK) + 1) * EN_CHUNK;
            if (q > g1) q = g1;
            work.push_back({p, q, i, 0});
            p = q;
        }
    }
    PCE_HIP(c, out_buf.reserve(sizeof(EnAcc) * (size_t)(n > 0 ? n : 1)));
    PCE_HIP(c, work_buf.reserve(sizeof(EnWork) * (work.size() + 1)));
    if (!work.empty())
        PCE_HIP(c, hipMemcpyAsync(work_buf.p, work.data(), sizeof(EnWork) * work.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));   // `work` is pageable and dies at return
    *n_work = (int64_t)work.size();
    return PCE_OK;
}

// Launch (async): zero the accumulators, stream the chunks.
int pce_energy_launch(pce_ctx *c, int32_t n, int32_t loud_thr, int64_t n_work, DevBuf &work_buf, DevBuf &out_buf, hipStream_t on)
{
    const hipStream_t st = on ? on : c->stream;
    PCE_HIP(c, hipMemsetAsync(out_buf.p, 0, sizeof(EnAcc) * (size_t)(n > 0 ? n : 1), st));
    if (n_work > 0) {
        KernelTimer t(c, PCE_K_ENERGY, st);
        // a batch beyond the 256 MB Infinity Cache is read with non-temporal loads (nothing re-reads a line in time: +11 % on a bare read loop)
        const bool nt = !c->clip_off.empty() && c->clip_off.back() * 2 > ((int64_t)256 << 20);
        // chunks per workgroup: as many as leave every CU >= 24 workgroups (7 are resident), at most 8 (the per-lane sums above).
        // Measured (us per launch at 82 MB / 400 MB / 1.6 GB): 1 chunk 21.5 / 73.5 / 310, 2: 23.2 / 68.9 / 296, 4: 23.2 / 72.9 / 288, 8: 30.0 / 79.1 / 291
        const int64_t cus = c->cu_count > 0 ? c->cu_count : 256;
        int cpb = c->en_cpb > 0 ? c->en_cpb : (int)std::min<int64_t>(8, std::max<int64_t>(1, n_work / (cus * 24)));
        if (cpb > 8) cpb = 8;
        const unsigned grid = (unsigned)((n_work + cpb - 1) / cpb);
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(EN_THREADS), 0, st, c->d_pcm, work_buf.as<EnWork>(), (int)n_work, cpb, (int)loud_thr, out_buf.as<EnAcc>());
        };
        if (nt) launch(k_energy<EN_LOADS, true>); else launch(k_energy<EN_LOADS, false>);
        PCE_HIP(c, hipGetLastError());
    }
    return PCE_OK;
}

// Device-side views of the accumulators for other translation units.
void pce_energy_range_ptrs(const DevBuf &out_buf, size_t *stride_bytes, const long long **sum, const int **m_hi, const int **m_lo)
{
    *stride_bytes = sizeof(EnAcc);
    const char *b = reinterpret_cast<const char *>(out_buf.p);
    *sum = reinterpret_cast<const long long *>(b + offsetof(EnAcc, sum));
    *m_hi = reinterpret_cast<const int *>(b + offsetof(EnAcc, m_hi));
    *m_lo = reinterpret_cast<const int *>(b + offsetof(EnAcc, m_lo));
}

const int *pce_energy_peak_ptr(const DevBuf &out_buf, size_t *stride_bytes)
{
    *stride_bytes = sizeof(EnAcc);
    return reinterpret_cast<const int *>(reinterpret_cast<const char *>(out_buf.p) + offsetof(EnAcc, peak));
}

extern "C" {

int pce_energy_run(pce_ctx *c, const pce_slice *slices, int32_t n, int32_t loud_thr)
{
    if (!c || (!slices && n > 0) || n < 0) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    PCE_HIP(c, hipSetDevice(c->device));
    if (c->lu_reads_en_out) {                                    // a LUFS chain on its side stream reads the peaks of the previous run out of en_out
        int rc = pce_side_join(c, pce_ctx::SIDE_LUFS); if (rc) return rc;
        c->lu_reads_en_out = false;
    }
    if (!c->en_cache.same(slices, n)) {
        c->en_n = -1;
        int st = pce_energy_plan(c, slices, n, c->en_work, c->en_out, &c->en_n_work);
        if (st) return st;
        c->en_cache.store(slices, n);
    }
    int st = pce_energy_launch(c, n, loud_thr, c->en_n_work, c->en_work, c->en_out, nullptr);
    if (st) return st;
    c->en_n = n;
    return PCE_OK;
}

int pce_energy_fetch(pce_ctx *c, pce_energy *out)
{
    if (!c || !out) return PCE_E_INVALID;
    if (c->en_n < 0) return pce_fail(c, PCE_E_STATE, "pce_energy_fetch before pce_energy_run");
    PCE_HIP(c, hipSetDevice(c->device));
    std::vector<EnAcc> acc((size_t)(c->en_n > 0 ? c->en_n : 1));
    if (c->en_n > 0)
        PCE_HIP(c, hipMemcpyAsync(acc.data(), c->en_out.p, sizeof(EnAcc) * (size_t)c->en_n, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    for (int32_t i = 0; i < c->en_n; i++) {
        const pce_slice &s = c->en_cache.v[(size_t)i];
        out[i].n = s.end - s.begin;
        out[i].sum_sq = (int64_t)acc[(size_t)i].sum_sq;
        out[i].sum_sq_wrap16 = (int64_t)acc[(size_t)i].sum_wrap;
        out[i].n_loud = (int64_t)acc[(size_t)i].n_loud;
        out[i].peak_abs = acc[(size_t)i].peak;
        out[i].reserved = 0;
    }
    return PCE_OK;
}

} // extern "C"

size_t pce_energy_stage_bytes(const pce_ctx *c) { return c->en_n > 0 ? sizeof(EnAcc) * (size_t)c->en_n : 0; }
int pce_energy_stage_enqueue(pce_ctx *c, void *pinned, std::vector<int64_t> &lens)
{
    lens.resize((size_t)(c->en_n > 0 ? c->en_n : 0));
    for (int32_t i = 0; i < c->en_n; i++) lens[(size_t)i] = c->en_cache.v[(size_t)i].end - c->en_cache.v[(size_t)i].begin;
    if (c->en_n > 0)
        PCE_HIP(c, hipMemcpyAsync(pinned, c->en_out.p, sizeof(EnAcc) * (size_t)c->en_n, hipMemcpyDeviceToHost, c->stream));
    return PCE_OK;
}
void pce_energy_stage_unpack(const void *pinned, const std::vector<int64_t> &lens, pce_energy *out)
{
    const EnAcc *acc = static_cast<const EnAcc *>(pinned);
    for (size_t i = 0; i < lens.size(); i++) {
        out[i].n = lens[i];
        out[i].sum_sq = (int64_t)acc[i].sum_sq;
        out[i].sum_sq_wrap16 = (int64_t)acc[i].sum_wrap;
        out[i].n_loud = (int64_t)acc[i].n_loud;
        out[i].peak_abs = acc[i].peak;
        out[i].reserved = 0;
    }
}
