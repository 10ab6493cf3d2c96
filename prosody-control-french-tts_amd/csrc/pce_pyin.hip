// pce_pyin.hip -- probabilistic YIN of the reference's viewers:
//   f0, voiced_flag, voiced_prob = librosa.pyin(audio, sr=sr, fmin=60, fmax=2000, hop_length=256)
//   (Code/visualisation/app.py:74-78, acoustic_analysis.py:76-94, visualisation_abtest/app.py:108-111).
// librosa is third party and absent from /root/reference: restated from its published implementation (pitch.py
// _cumulative_mean_normalized_difference / _parabolic_interpolation / __pyin_helper, sequence.py transition_local /
// viterbi), parity unpinned; oracle/pyin_oracle.py is the numpy restatement the tests compare with.
//
// k_pyin_frames  one wavefront per frame.  The 2048-sample frame (centred, zero padded) is staged in LDS as doubles
//                holding the int16 VALUES: the cross-correlation sum_{j=1..1024} x[j] x[j+tau] and the window energies are then
//                exact integer arithmetic in fp64 (< 2^41), as are (after librosa's 1e-6 clamps) the difference function
//                and its cumulative sums: the cumulative-mean normalised difference equals the float64 evaluation of
//                librosa's expressions bit for bit (its float32 FFT / cumsum route carries rounding noise around it).
//                Lane = lag (tau = lane + 64 g), one fp64 FMA per term.  Then, still per wavefront: troughs, the 100-threshold Boltzmann/beta trough probabilities (lane =
//                trough, ballot/popcount ranks), parabolic refinement, pitch-bin observation list, voiced probability.
// k_pyin_viterbi one workgroup per clip: the 2 x n_bins state HMM, exactly librosa's dense argmax (first maximum):
//                in-band predecessors from the transition table, out-of-band ones (log(0 + tiny)) through prefix/suffix
//                maxima of the previous column; back-pointers in HBM, back-tracking by one lane.
// Host-computed tables (numpy, so that every constant has the bits the restatement uses): thresholds, beta
// probabilities and their prefix sums, Boltzmann factors, log transition rows.
#include "pce_internal.h"

namespace {

constexpr int PY_FRAME = 2048, PY_WIN = 1024, PY_MAXTAU = 1024, PY_G = 16;      // lags per lane: tau = lane + 64 g
constexpr int PY_MAXTR = 512;            // troughs per frame: local minima are at least 2 lags apart, n_tau <= 1016
constexpr int PY_WPB = 2;                // frames (waves) per workgroup

struct PyPlan {
    int hop, min_period, max_period, n_tau, n_bins, half, n_thr, n_groups;
    double sr, fmin, bins_per_octave, no_trough_prob, c0 /* log(tiny) */, log_pinit, tiny;
    // offsets (doubles) into the table blob
    int o_thr, o_beta, o_bprefix, o_bfac, o_bexp, o_lt_same, o_lt_sw;      // log transition table: [e + half][k][same | switch], k = predecessor bin
};
struct PyObs { int n; int status; double voiced_prob; double log_unvoiced; };   // per frame header
// per frame: header + PY_MAXTR (bin, log-prob) pairs

__device__ __forceinline__ double shfl_up_f64(double v, int d)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, d, 64); hi = __shfl_up(hi, d, 64);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shfl_f64(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src, 64); hi = __shfl(hi, src, 64);
    return __hiloint2double(hi, lo);
}

template <int NG>   // lag groups per lane (5 at 16 kHz, 12 at 44.1 kHz); 0: taken from the plan at run time
__global__ __launch_bounds__(64 * PY_WPB) void k_pyin_frames(const int16_t *__restrict__ pcm, const int64_t *__restrict__ clip_off,
                                                           const int64_t *__restrict__ frame_off, int n_clips, PyPlan P,
                                                           const double *__restrict__ tab, PyObs *__restrict__ hdr,
                                                           short *__restrict__ obs_bin, double *__restrict__ obs_lp)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // per wave: x[2048 + 64] | y[PY_MAXTAU + 2] | trh[PY_MAXTR] | tri[PY_MAXTR] (ints in doubles' storage)
    double *x = lds + (size_t)wv * (PY_FRAME + 64 + PY_MAXTAU + 2 + 2 * PY_MAXTR);
    double *y = x + PY_FRAME + 64;
    double *trh = y + PY_MAXTAU + 2;
    int *tri = reinterpret_cast<int *>(trh + PY_MAXTR);
    const int clip = blockIdx.y;
    const int64_t c0 = clip_off[clip], len = clip_off[clip + 1] - c0;
    const int64_t f0 = frame_off[clip], nf = frame_off[clip + 1] - f0;
    for (int64_t fr = (int64_t)blockIdx.x * PY_WPB + wv; fr < nf; fr += (int64_t)gridDim.x * PY_WPB) {
        // ---- stage the frame: padded index i <-> sample fr * hop - 1024 + i
        const int64_t s0 = fr * P.hop - PY_FRAME / 2;
        for (int i = lane; i < PY_FRAME + 64; i += 64) {
            const int64_t s = s0 + i;
            x[i] = (i < PY_FRAME && s >= 0 && s < len) ? (double)pcm[c0 + s] : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- cross-correlation, exact: acf[tau] = sum_{j=1..1024} x[j] x[j + tau], tau = lane + 64 g <= max_period
        double d[PY_G];
#pragma unroll
        for (int g = 0; g < PY_G; g++) d[g] = 0.0;
        const int ng = NG > 0 ? NG : P.n_groups;                    // groups that hold a lag <= max_period
        for (int j = 1; j <= PY_WIN; j++) {
            const double xj = x[j];
#pragma unroll
            for (int g = 0; g < PY_G; g++)
                if (g < ng) d[g] = fma(xj, x[j + lane + 64 * g], d[g]);       // j + tau <= 1024 + 63 + 64 (ng - 1) < 2112
        }
        // ---- window energies E[tau] = sum_{j=tau+1..tau+1024} x[j]^2 = E[0] + sum_{u<=tau} (x[u+1024]^2 - x[u]^2), exact
        double e0 = 0.0;
        for (int j = 1 + lane; j <= PY_WIN; j += 64) e0 = fma(x[j], x[j], e0);
        for (int o = 32; o > 0; o >>= 1) e0 += shfl_f64(e0, lane ^ o);
        const double sc = 9.31322574615478515625e-10;               // 2^-30: int16 values -> [-1, 1) samples, squared
        auto clamp6 = [](double v) { return fabs(v) < 1e-6 ? 0.0 : v; };      // librosa: acf / energy below 1e-6 are set to 0
        const double E0 = clamp6(e0 * sc);
        // ---- difference function d = E[0] + E[tau] - 2 acf[tau], its cumulative sums over tau = 1.., the normalised
        //      difference on [min_period, max_period]; every value is a multiple of 2^-30 below 2^22: exact
        double ecarry = 0.0, carry = 0.0;
#pragma unroll
        for (int g = 0; g < PY_G; g++) {
            if (g < ng) {
                const int tau = lane + 64 * g;
                const bool in = tau >= 1 && tau <= P.max_period;
                const double hi = x[tau + PY_WIN], lo = x[tau];
                double de = in ? hi * hi - lo * lo : 0.0;
                for (int o = 1; o < 64; o <<= 1) { const double u = shfl_up_f64(de, o); if (lane >= o) de += u; }
                const double Et = clamp6((e0 + (ecarry + de)) * sc);
                ecarry += shfl_f64(de, 63);
                const double v = in ? (E0 + Et) - 2.0 * clamp6(d[g] * sc) : 0.0;
                double s = v;
                for (int o = 1; o < 64; o <<= 1) { const double u = shfl_up_f64(s, o); if (lane >= o) s += u; }
                const double cum = carry + s;
                carry += shfl_f64(s, 63);
                if (tau >= P.min_period && tau <= P.max_period) y[tau - P.min_period] = v / (cum / (double)tau + P.tiny);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- troughs (librosa.util.localmin with edge padding; the first element by its own rule), compacted in lag order
        const int n = P.n_tau;
        int m_total = 0;
        for (int b0 = 0; b0 < n; b0 += 64) {
            const int i = b0 + lane;
            bool tr = false;
            if (i < n) {
                const double yi = y[i];
                if (i == 0) tr = yi < y[1];
                else tr = (yi < y[i - 1]) && (yi <= (i + 1 < n ? y[i + 1] : yi));
            }
            const unsigned long long bal = __ballot(tr);
            if (tr) {
                const int pos = m_total + __popcll(bal & ((1ull << lane) - 1ull));
                if (pos < PY_MAXTR) { trh[pos] = y[i]; tri[pos] = i; }
            }
            m_total += __popcll(bal);
        }
        const int status = m_total > PY_MAXTR ? 1 : 0;
        const int M = m_total > PY_MAXTR ? PY_MAXTR : m_total;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- trough probabilities: lane = trough (m = lane + 64 r), loop over the thresholds
        constexpr int R = PY_MAXTR / 64;
        double h[R], pr[R];
#pragma unroll
        for (int r = 0; r < R; r++) { const int m = lane + 64 * r; h[r] = m < M ? trh[m] : 1e300; pr[r] = 0.0; }
        const double *thr = tab + P.o_thr, *beta = tab + P.o_beta, *bfac = tab + P.o_bfac, *bexp = tab + P.o_bexp;
        for (int k = 0; k < P.n_thr; k++) {
            const double th = thr[k];
            unsigned long long bal[R]; int before = 0, total = 0;
#pragma unroll
            for (int r = 0; r < R; r++) { bal[r] = __ballot(h[r] < th); total += __popcll(bal[r]); }
            if (total == 0) continue;
            const double f = bfac[total] * beta[k];
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (h[r] < th) {
                    const int rank = before + __popcll(bal[r] & ((1ull << lane) - 1ull));
                    pr[r] += (bfac[total] * bexp[rank]) * beta[k];
                }
                before += __popcll(bal[r]);
            }
            (void)f;
        }
        // global minimum trough (first one) takes the no-trough mass of the thresholds at or below its height
        if (M > 0) {
            double best = 1e300; int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < R; r++) { const int m = lane + 64 * r; if (m < M && (h[r] < best)) { best = h[r]; bi = m; } }
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = shfl_f64(best, lane ^ o); const int oi = __shfl(bi, lane ^ o, 64);
                if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            int nb = 0;                                             // thresholds NOT above the minimum: !(h < thr[k])
            for (int k = lane; k < P.n_thr; k += 64) nb += !(best < thr[k]);
            for (int o = 32; o > 0; o >>= 1) nb += __shfl(nb, lane ^ o, 64);
            const double extra = P.no_trough_prob * tab[P.o_bprefix + nb];
#pragma unroll
            for (int r = 0; r < R; r++) if (lane + 64 * r == bi) pr[r] += extra;
        }
        // ---- candidates: nonzero probabilities, parabolic refinement, pitch bin; a later trough in the same bin wins
        PyObs *H = hdr + f0 + fr;
        short *ob = obs_bin + (size_t)(f0 + fr) * PY_MAXTR;
        double *ol = obs_lp + (size_t)(f0 + fr) * PY_MAXTR;
        int bins[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int m = lane + 64 * r;
            bins[r] = -1;
            if (m < M && pr[r] != 0.0) {
                const int i = tri[m];
                double shift = 0.0;
                if (i >= 1 && i + 1 < n) {
                    const double ym = y[i - 1], y0 = y[i], yp = y[i + 1];
                    const double a = yp + ym - 2.0 * y0, b = (yp - ym) / 2.0;
                    if (!(fabs(b) >= fabs(a))) shift = -b / a;
                }
                const double period = (double)(P.min_period + i) + shift;
                const double fc = P.sr / period;
                double bi = rint(P.bins_per_octave * log2(fc / P.fmin));
                if (!(bi >= 0.0)) bi = 0.0;
                if (bi > (double)P.n_bins) bi = (double)P.n_bins;
                bins[r] = (int)bi;
            }
        }
        // compact the nonzero ones in trough order; drop an entry whose successor has the same bin, and bin == n_bins
        // (librosa writes that one into the first unvoiced row, which is overwritten afterwards)
        int cnt = 0;
        double vp = 0.0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool nz = bins[r] >= 0;
            const unsigned long long bal = __ballot(nz);
            if (nz) { const int pos = cnt + __popcll(bal & ((1ull << lane) - 1ull)); tri[pos] = bins[r]; trh[pos] = pr[r]; }
            cnt += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int kept = 0;
        for (int b0 = 0; b0 < cnt; b0 += 64) {
            const int q = b0 + lane;
            bool keep = false; int bq = 0; double pq = 0.0;
            if (q < cnt) {
                bq = tri[q]; pq = trh[q];
                keep = bq < P.n_bins && !(q + 1 < cnt && tri[q + 1] == bq);
            }
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                const int pos = kept + __popcll(bal & ((1ull << lane) - 1ull));
                ob[pos] = (short)bq; ol[pos] = log(pq + P.tiny);
                vp += pq;
            }
            kept += __popcll(bal);
        }
        for (int o = 32; o > 0; o >>= 1) vp += shfl_f64(vp, lane ^ o);
        if (vp > 1.0) vp = 1.0;
        if (vp < 0.0) vp = 0.0;
        if (lane == 0) {
            H->n = kept; H->status = status; H->voiced_prob = vp;
            H->log_unvoiced = log((1.0 - vp) / (double)P.n_bins + P.tiny);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// HMM decoding.  State s = v * n_bins + j (v = 0 voiced, 1 unvoiced).  Thread j owns states (0, j) and (1, j).
// ---------------------------------------------------------------------------------------------------------------
constexpr int VT_THREADS = 640;

struct MaxIdx { double v; int i; };
__device__ __forceinline__ MaxIdx better_first(MaxIdx a, MaxIdx b)     // maximum, the smaller index on ties
{
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__global__ __launch_bounds__(VT_THREADS) void k_pyin_viterbi(const int64_t *__restrict__ frame_off, PyPlan P, const double *__restrict__ tab,
                                                            const PyObs *__restrict__ hdr, const short *__restrict__ obs_bin,
                                                            const double *__restrict__ obs_lp, unsigned short *__restrict__ ptr,
                                                            int *__restrict__ states)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int nb = P.n_bins, half = P.half;
    double *V = lds;                       // [2][nb] previous column
    double *lo = V + 2 * nb;               // [nb] log observation of the voiced states of the current frame
    double *pmv = lo + nb;                 // prefix maxima value [2][nb]
    double *smv = pmv + 2 * nb;            // suffix maxima value [2][nb]
    int *pmi = reinterpret_cast<int *>(smv + 2 * nb);   // [2][nb]
    int *smi = pmi + 2 * nb;               // [2][nb]
    __shared__ MaxIdx wtot[2][VT_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, j = tid;
    const int clip = blockIdx.x;
    const int64_t f0 = frame_off[clip], T = frame_off[clip + 1] - f0;
    if (T <= 0) return;
    // log T[k -> j] for predecessor k (bin) and e = j - k in [-half, half].  librosa normalises every row by its own
    // sum over all n_bins entries, whose rounding differs from row to row by an ulp: the table keeps every row.
    // (same-voicing, switching) pairs interleaved [e + half][k][2]: one 16-byte load per predecessor
    const double2 *lt2 = reinterpret_cast<const double2 *>(tab + P.o_lt_same);
    unsigned short *pt = ptr + (size_t)f0 * (size_t)(2 * nb);
    for (int64_t t = 0; t < T; t++) {
        // ---- observation column
        const PyObs H = hdr[f0 + t];
        if (j < nb) lo[j] = P.c0;
        __syncthreads();
        if (tid < H.n) lo[obs_bin[(size_t)(f0 + t) * PY_MAXTR + tid]] = obs_lp[(size_t)(f0 + t) * PY_MAXTR + tid];
        __syncthreads();
        double nv0 = 0.0, nv1 = 0.0;
        if (t == 0) {
            if (j < nb) { nv0 = lo[j] + P.log_pinit; nv1 = H.log_unvoiced + P.log_pinit; }
        } else {
            // ---- prefix / suffix maxima (first index on ties) of both halves of the previous column
#pragma unroll
            for (int v = 0; v < 2; v++) {
                MaxIdx a; a.v = j < nb ? V[v * nb + j] : -1e308; a.i = j < nb ? j : 0x7fffffff;
                MaxIdx p = a;
                for (int o = 1; o < 64; o <<= 1) {
                    MaxIdx q; q.v = shfl_up_f64(p.v, o); q.i = __shfl_up(p.i, o, 64);
                    if (lane >= o) p = better_first(p, q);
                }
                if (lane == 63) wtot[v][wv] = p;
                if (j < nb) { pmv[v * nb + j] = p.v; pmi[v * nb + j] = p.i; }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 2; v++) {
                if (j < nb && wv > 0) {
                    MaxIdx p; p.v = pmv[v * nb + j]; p.i = pmi[v * nb + j];
                    for (int w = 0; w < wv; w++) p = better_first(p, wtot[v][w]);
                    pmv[v * nb + j] = p.v; pmi[v * nb + j] = p.i;
                }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 2; v++) {                          // suffix: scan the reversed order
                const int jr = nb - 1 - j;                          // this thread's element in reversed order
                MaxIdx a; a.v = j < nb ? V[v * nb + jr] : -1e308; a.i = j < nb ? jr : 0x7fffffff;
                MaxIdx p = a;
                for (int o = 1; o < 64; o <<= 1) {
                    MaxIdx q; q.v = shfl_up_f64(p.v, o); q.i = __shfl_up(p.i, o, 64);
                    if (lane >= o) p = better_first(p, q);
                }
                if (lane == 63) wtot[v][wv] = p;
                if (j < nb) { smv[v * nb + jr] = p.v; smi[v * nb + jr] = p.i; }
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 2; v++) {
                const int jr = nb - 1 - j;
                if (j < nb && wv > 0) {
                    MaxIdx p; p.v = smv[v * nb + jr]; p.i = smi[v * nb + jr];
                    for (int w = 0; w < wv; w++) p = better_first(p, wtot[v][w]);
                    smv[v * nb + jr] = p.v; smi[v * nb + jr] = p.i;
                }
            }
            __syncthreads();
            if (j < nb) {
                // candidates in index order, strict '>' keeps the first maximum (np.argmax)
                double b0 = -1e308, b1 = -1e308; int i0 = 0, i1 = 0;     // best for target (0, j) and (1, j)
                const int klo = j - half > 0 ? j - half : 0, khi = j + half < nb - 1 ? j + half : nb - 1;
#pragma unroll
                for (int v = 0; v < 2; v++) {                      // predecessor half v
                    const double *Vv = V + v * nb;
                    if (klo > 0) {                                  // out of band below: log(0 + tiny)
                        const double c = pmv[v * nb + klo - 1] + P.c0; const int ci = v * nb + pmi[v * nb + klo - 1];
                        if (c > b0) { b0 = c; i0 = ci; }
                        if (c > b1) { b1 = c; i1 = ci; }
                    }
                    // ascending k (= descending e = j - k), fixed trip count with a validity test.  29 us per frame (18 ms per
                    // 626-frame clip, one CU per clip): 284 table loads per state and frame (the 690 KB table fits neither
                    // LDS nor registers).  Sending the interior rows as a base row + int8 bit-pattern differences (an eighth
                    // of the bytes, same number of load instructions) measured 2.6x SLOWER: the loop is bound by load
                    // instructions, not bytes.
#pragma unroll 8
                    for (int e = half; e >= -half; e--) {
                        const int k = j - e;
                        const bool ok = k >= 0 && k < nb;
                        const int kk = ok ? k : j;
                        const double pv = Vv[kk];
                        const double2 tt = lt2[((ok ? e : 0) + half) * nb + kk];
                        const double same = tt.x, sw = tt.y;
                        const double c0v = pv + (v == 0 ? same : sw), c1v = pv + (v == 0 ? sw : same);
                        if (ok && c0v > b0) { b0 = c0v; i0 = v * nb + k; }
                        if (ok && c1v > b1) { b1 = c1v; i1 = v * nb + k; }
                    }
                    if (khi < nb - 1) {
                        const double c = smv[v * nb + khi + 1] + P.c0; const int ci = v * nb + smi[v * nb + khi + 1];
                        if (c > b0) { b0 = c; i0 = ci; }
                        if (c > b1) { b1 = c; i1 = ci; }
                    }
                }
                nv0 = lo[j] + b0; nv1 = H.log_unvoiced + b1;
                pt[(size_t)t * (2 * nb) + j] = (unsigned short)i0;
                pt[(size_t)t * (2 * nb) + nb + j] = (unsigned short)i1;
            }
        }
        __syncthreads();
        if (j < nb) { V[j] = nv0; V[nb + j] = nv1; }
        __syncthreads();
    }
    // ---- last state (first maximum) and back-tracking
    if (tid == 0) {
        double best = V[0]; int bi = 0;
        for (int s = 1; s < 2 * nb; s++) if (V[s] > best) { best = V[s]; bi = s; }
        int *st = states + f0;
        st[T - 1] = bi;
        for (int64_t t = T - 2; t >= 0; t--) { bi = pt[(size_t)(t + 1) * (2 * nb) + bi]; st[t] = bi; }
    }
}

} // namespace

extern "C" {

int pce_pyin_run(pce_ctx *c, const pce_pyin_plan *plan, const double *tables, int64_t n_tables)
{
    if (!c || !plan || !tables) return PCE_E_INVALID;
    if (!c->d_pcm) return pce_fail(c, PCE_E_STATE, "no batch uploaded");
    PyPlan P{};
    P.hop = plan->hop_length; P.min_period = plan->min_period; P.max_period = plan->max_period;
    P.n_tau = P.max_period - P.min_period + 1; P.n_bins = plan->n_pitch_bins; P.half = plan->trans_width / 2; P.n_thr = plan->n_thresholds;
    P.sr = plan->sr; P.fmin = plan->fmin; P.bins_per_octave = plan->bins_per_octave; P.no_trough_prob = plan->no_trough_prob;
    P.c0 = plan->log_tiny; P.log_pinit = plan->log_p_init; P.tiny = plan->tiny;
    P.n_groups = P.max_period / 64 + 1;
    if (plan->frame_length != PY_FRAME || P.hop < 1 || P.min_period < 1 || P.max_period >= PY_MAXTAU || P.max_period <= P.min_period + 2 ||
        P.n_groups > PY_G || P.n_bins < 2 * P.half + 2 || P.n_bins > 640 || !(plan->trans_width & 1) || P.n_thr < 1 || P.n_thr > 128)
        return pce_fail(c, PCE_E_LIMIT, "unsupported pYIN plan (frame_length 2048, max_period < 1024, n_pitch_bins <= 640, odd transition width)");
    const int W = 2 * P.half + 1;
    P.o_thr = 0; P.o_beta = P.o_thr + P.n_thr; P.o_bprefix = P.o_beta + P.n_thr; P.o_bfac = P.o_bprefix + P.n_thr + 1;
    P.o_bexp = P.o_bfac + PY_MAXTR + 1; P.o_lt_same = (P.o_bexp + PY_MAXTR + 1 + 1) & ~1; P.o_lt_sw = P.o_lt_same + W * P.n_bins;   // (16-byte aligned pairs)
    const int64_t expect = (int64_t)P.o_lt_sw + (int64_t)W * P.n_bins;
    if (n_tables != expect) return pce_fail(c, PCE_E_INVALID, "pYIN table blob has %lld doubles, expected %lld", (long long)n_tables, (long long)expect);
    PCE_HIP(c, hipSetDevice(c->device));
    c->py_ran = false;
    c->py_off.assign((size_t)c->n_clips + 1, 0);
    int64_t max_frames = 0;
    for (int32_t i = 0; i < c->n_clips; i++) {
        const int64_t len = c->clip_off[(size_t)i + 1] - c->clip_off[(size_t)i];
        const int64_t nf = 1 + len / P.hop;
        c->py_off[(size_t)i + 1] = c->py_off[(size_t)i] + nf;
        if (nf > max_frames) max_frames = nf;
    }
    const int64_t total = c->py_off[(size_t)c->n_clips];
    if (c->n_clips == 0) { c->py_ran = true; return PCE_OK; }
    PCE_HIP(c, c->py_doff.reserve(sizeof(int64_t) * ((size_t)c->n_clips + 1)));
    PCE_HIP(c, c->py_tab.reserve(sizeof(double) * (size_t)n_tables));
    PCE_HIP(c, c->py_hdr.reserve(sizeof(PyObs) * (size_t)total));
    PCE_HIP(c, c->py_bin.reserve(sizeof(short) * (size_t)total * PY_MAXTR));
    PCE_HIP(c, c->py_lp.reserve(sizeof(double) * (size_t)total * PY_MAXTR));
    PCE_HIP(c, c->py_ptr.reserve(sizeof(unsigned short) * (size_t)total * (size_t)(2 * P.n_bins)));
    PCE_HIP(c, c->py_states.reserve(sizeof(int) * (size_t)total));
    PCE_HIP(c, hipMemcpyAsync(c->py_doff.p, c->py_off.data(), sizeof(int64_t) * ((size_t)c->n_clips + 1), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(c->py_tab.p, tables, sizeof(double) * (size_t)n_tables, hipMemcpyHostToDevice, c->stream));
    {
        const size_t lds = sizeof(double) * (size_t)PY_WPB * (PY_FRAME + 64 + PY_MAXTAU + 2 + 2 * PY_MAXTR);
        const void *kfn = P.n_groups == 5 ? reinterpret_cast<const void *>(k_pyin_frames<5>)
                          : P.n_groups == 12 ? reinterpret_cast<const void *>(k_pyin_frames<12>) : reinterpret_cast<const void *>(k_pyin_frames<0>);
        PCE_HIP(c, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int64_t gx = (max_frames + PY_WPB - 1) / PY_WPB; if (gx > 2048) gx = 2048;
        const unsigned gy = (unsigned)(c->n_clips < 65535 ? c->n_clips : 65535);
        if (c->n_clips > 65535) return pce_fail(c, PCE_E_LIMIT, "pYIN: more than 65535 clips in one batch");
        KernelTimer t(c, PCE_K_PYIN_FRAMES);
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)gx, gy), dim3(64 * PY_WPB), lds, c->stream, c->d_pcm, c->d_clip_off.as<int64_t>(),
                               c->py_doff.as<int64_t>(), (int)c->n_clips, P, c->py_tab.as<double>(), c->py_hdr.as<PyObs>(), c->py_bin.as<short>(),
                               c->py_lp.as<double>());
        };
        if (P.n_groups == 5) launch(k_pyin_frames<5>);
        else if (P.n_groups == 12) launch(k_pyin_frames<12>);
        else launch(k_pyin_frames<0>);
    }
    {
        const size_t lds = sizeof(double) * (size_t)(2 * P.n_bins + P.n_bins + 4 * P.n_bins) + sizeof(int) * (size_t)(4 * P.n_bins) + 64;
        PCE_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_pyin_viterbi), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KernelTimer t(c, PCE_K_PYIN_VITERBI);
        hipLaunchKernelGGL(k_pyin_viterbi, dim3((unsigned)c->n_clips), dim3(VT_THREADS), lds, c->stream, c->py_doff.as<int64_t>(), P,
                           c->py_tab.as<double>(), c->py_hdr.as<PyObs>(), c->py_bin.as<short>(), c->py_lp.as<double>(),
                           c->py_ptr.as<unsigned short>(), c->py_states.as<int>());
    }
    PCE_HIP(c, hipGetLastError());
    PCE_HIP(c, hipStreamSynchronize(c->stream));                 // `tables` and py_off were sources of asynchronous copies
    c->py_ran = true;
    return PCE_OK;
}

int pce_pyin_shape(pce_ctx *c, int32_t clip, int64_t *n_frames)
{
    if (!c || !n_frames) return PCE_E_INVALID;
    if (!c->py_ran) return pce_fail(c, PCE_E_STATE, "pce_pyin_shape before pce_pyin_run");
    if (clip < 0 || clip >= c->n_clips) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    *n_frames = c->py_off[(size_t)clip + 1] - c->py_off[(size_t)clip];
    return PCE_OK;
}

int pce_pyin_fetch(pce_ctx *c, int32_t clip, int32_t *states, double *voiced_prob, int32_t *status)
{
    if (!c) return PCE_E_INVALID;
    if (!c->py_ran) return pce_fail(c, PCE_E_STATE, "pce_pyin_fetch before pce_pyin_run");
    if (clip < 0 || clip >= c->n_clips) return pce_fail(c, PCE_E_INVALID, "clip out of range");
    PCE_HIP(c, hipSetDevice(c->device));
    const int64_t f0 = c->py_off[(size_t)clip], nf = c->py_off[(size_t)clip + 1] - f0;
    std::vector<PyObs> h((size_t)nf);
    if (states) PCE_HIP(c, hipMemcpyAsync(states, c->py_states.as<int>() + f0, sizeof(int) * (size_t)nf, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipMemcpyAsync(h.data(), c->py_hdr.as<PyObs>() + f0, sizeof(PyObs) * (size_t)nf, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    int st = 0;
    for (int64_t i = 0; i < nf; i++) { if (voiced_prob) voiced_prob[i] = h[(size_t)i].voiced_prob; st |= h[(size_t)i].status; }
    if (status) *status = st;
    return PCE_OK;
}

} // extern "C"
