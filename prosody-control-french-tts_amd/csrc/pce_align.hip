// pce_align.hip -- dynamic time warping of token x frame cost matrices (R8 alignment step).
//
// Replaces the DTW that turns Whisper cross-attention into word timestamps
// (Code/Aligners/use_whisper_timestamped.py:163 -> whisper_timestamped / openai-whisper timing.py,
// third-party): cost[i+1][j+1] = x[i][j] + min(cost[i][j], cost[i][j+1], cost[i+1][j]) with the
// trace preference diagonal, up, left exactly as openai-whisper's dtw_cpu writes it
// ("c0 < c1 and c0 < c2 -> 0; elif c1 < c0 and c1 < c2 -> 1; else 2"), boundary trace[0,:] = 2,
// trace[:,0] = 1, back-tracking from (N, M).  fp64 adds and compares only, so the path indices are
// bit-identical to the CPU recurrence on the same matrix.
//
// One workgroup per matrix, thread = row, the anti-diagonals i + j = const sweep the matrix; the two
// previous diagonals live in LDS (double buffered), the 1-byte trace goes to global memory and is walked
// back by one lane.  N <= 1024 rows (Whisper: <= 448 tokens), any number of columns.
#include "pce_internal.h"
#include <vector>

namespace {

constexpr int DTW_MAXN = 1024;

// matrix b: N = n_rows[b] (or n_rows_all), M = n_cols[b], element (i, j) at x[b * x_stride + i * ld + j];
// trace / path buffers are strided by the batch maxima (N_max, M_max).
__global__ __launch_bounds__(DTW_MAXN) void k_dtw(const double *__restrict__ x, int64_t x_stride, int ld, const int *__restrict__ n_rows,
                                                 const int *__restrict__ n_cols, int N_max, int M_max, unsigned char *__restrict__ trace,
                                                 int *__restrict__ path_i, int *__restrict__ path_j, int *__restrict__ path_len)
{
    // cost on diagonals d-2, d-1, d (index = row of the padded matrix), kept in float32 as openai-whisper's dtw_cpu keeps
    // its cost array (np.float32): every cell is the float64 sum x + c rounded to float32, comparisons are on float32
    __shared__ float diag[3][DTW_MAXN + 1];
    const int b = blockIdx.x, t = threadIdx.x;
    const int N = n_rows ? n_rows[b] : N_max, M = n_cols ? n_cols[b] : M_max;
    if (N <= 0 || M <= 0) { if (t == 0) path_len[b] = 0; return; }
    const double *xb = x + (size_t)b * (size_t)x_stride;
    unsigned char *tr = trace + (size_t)b * (size_t)(N_max + 1) * (size_t)(M_max + 1);
    const float INF = __builtin_huge_valf();
    // padded cost matrix C[(N+1) x (M+1)]: C[0][0] = 0, rest of row 0 / column 0 = inf.
    // diagonal D (of the padded matrix) holds C[r][D - r]; thread t owns padded row r = t + 1.
    for (int r = t; r <= N; r += blockDim.x) { diag[0][r] = INF; diag[1][r] = INF; diag[2][r] = INF; }
    __syncthreads();
    if (t == 0) diag[0][0] = 0.0f;                // diagonal 0: C[0][0]
    // diagonal 1: C[0][1] = inf, C[1][0] = inf (already inf)
    __syncthreads();
    const int r = t + 1;
    for (int D = 2; D <= N + M; D++) {
        float *cur = diag[D % 3]; const float *p1 = diag[(D - 1) % 3], *p2 = diag[(D - 2) % 3];
        const int c = D - r;                      // padded column
        if (t < N && c >= 1 && c <= M) {
            const float c0 = p2[r - 1], c1 = p1[r - 1], c2 = p1[r];       // C[r-1][c-1], C[r-1][c], C[r][c-1]
            float cm; unsigned char tt;
            if (c0 < c1 && c0 < c2) { cm = c0; tt = 0; }
            else if (c1 < c0 && c1 < c2) { cm = c1; tt = 1; }
            else { cm = c2; tt = 2; }
            cur[r] = (float)(xb[(size_t)(r - 1) * ld + (c - 1)] + (double)cm);
            tr[(size_t)r * (M + 1) + c] = tt;
        } else if (t < N) {
            cur[r] = INF;
        }
        if (t == 0) cur[0] = INF;                 // C[0][D] = inf for D >= 1
        __syncthreads();
    }
    // back-trace (one lane): trace[0][:] = 2, trace[:][0] = 1
    __threadfence_block();
    if (t == 0) {
        int i = N, j = M, n = 0;
        int *pi = path_i + (size_t)b * (N_max + M_max), *pj = path_j + (size_t)b * (N_max + M_max);
        while (i > 0 || j > 0) {
            pi[n] = i - 1; pj[n] = j - 1; n++;
            const int tt = (i == 0) ? 2 : (j == 0) ? 1 : tr[(size_t)i * (M + 1) + j];
            if (tt == 0) { i--; j--; } else if (tt == 1) { i--; } else { j--; }
        }
        // reverse in place
        for (int a = 0, z = n - 1; a < z; a++, z--) {
            const int ti = pi[a], tj = pj[a]; pi[a] = pi[z]; pj[a] = pj[z]; pi[z] = ti; pj[z] = tj;
        }
        path_len[b] = n;
    }
}

// ---------------------------------------------------------------------------
// Batched Needleman-Wunsch word alignment (legacy Pipeline, Code/Pipeline/NeedlemanWunschAlignement.py:27-81):
// score[i][j] = max(score[i-1][j-1] + (a_i == b_j ? match : mismatch), score[i-1][j] + gap, score[i][j-1] + gap),
// score[i][0] = i gap, score[0][j] = j gap; the trace-back tests "diagonal, then up, else left" (:69-80) against
// the finished matrix, which is the same as recording, per cell, the FIRST of (diagonal, up, left) that attains the
// maximum.  Tokens are compared on the host-normalised integer ids.  Integer arithmetic: the alignment is exactly
// the reference's.  Same wavefront as k_dtw: one workgroup per pair, thread = row, two previous anti-diagonals in LDS; more
// than 1 024 rows are swept in stripes of 1 024 (no length limit).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(DTW_MAXN) void k_nw(const int *__restrict__ a_ids, const long long *__restrict__ a_off, const int *__restrict__ b_ids,
                                                const long long *__restrict__ b_off, int match, int mismatch, int gap,
                                                unsigned char *__restrict__ trace, const long long *__restrict__ tr_off,
                                                int *__restrict__ out_i, int *__restrict__ out_j, const long long *__restrict__ out_off,
                                                int *__restrict__ out_len, int *__restrict__ rows, const long long *__restrict__ rows_off)
{
    __shared__ int diag[3][DTW_MAXN + 1];
    const int b = blockIdx.x, t = threadIdx.x, S = (int)blockDim.x;
    const int N = (int)(a_off[b + 1] - a_off[b]), M = (int)(b_off[b + 1] - b_off[b]);
    const int *A = a_ids + a_off[b], *B = b_ids + b_off[b];
    unsigned char *tr = trace + tr_off[b];                // [(N+1) x (M+1)]
    // Rows are swept in stripes of blockDim.x (round 5: any number of rows; one stripe is the old kernel).  Inside a stripe, local diagonal
    // D holds score[r0 + lr][D - lr] at index lr; thread t owns local row lr = t + 1, index 0 is the row above the stripe: the boundary
    // row score[0][c] = c gap for the first stripe, the previous stripe's last row (handed over through `top`, global) for the others.
    int *top = rows + rows_off[b], *bot = top + (M + 1);  // only touched when N > blockDim.x
    for (int r0 = 0; r0 < N; r0 += S) {
        const int R = min(S, N - r0), lr = t + 1, gr = r0 + lr;
        const bool first = r0 == 0, more = r0 + S < N;
        const int my_a = t < R ? A[gr - 1] : 0;
        if (t == 0) diag[0][0] = r0 * gap;                // score[r0][0]
        int tv = (t == 0 && M >= 1) ? (first ? gap : top[1]) : 0;         // score[r0][D] for the coming diagonal, fetched one step ahead
        __syncthreads();
        for (int D = 1; D <= R + M; D++) {
            int *cur = diag[D % 3]; const int *p1 = diag[(D + 2) % 3], *p2 = diag[(D + 1) % 3];
            const int c = D - lr;
            if (t < R) {
                if (c >= 1 && c <= M) {
                    const int dg = p2[lr - 1] + (my_a == B[c - 1] ? match : mismatch), up = p1[lr - 1] + gap, lf = p1[lr] + gap;
                    int best = dg; unsigned char tt = 0;
                    if (up > best) { best = up; tt = 1; }
                    if (lf > best) { best = lf; tt = 2; }
                    cur[lr] = best;
                    tr[(size_t)gr * (M + 1) + c] = tt;
                    if (more && lr == R) bot[c] = best;   // the stripe's last row: the next stripe's boundary
                } else if (c == 0) {
                    cur[lr] = gr * gap;                   // score[gr][0]
                }
            }
            if (t == 0 && D <= M) { cur[0] = tv; if (D + 1 <= M) tv = first ? (D + 1) * gap : top[D + 1]; }      // score[r0][D]
            __syncthreads();
        }
        if (more) { __threadfence(); __syncthreads(); int *x = top; top = bot; bot = x; }
    }
    __threadfence_block();
    if (t == 0) {
        int i = N, j = M, n = 0;
        int *pi = out_i + out_off[b], *pj = out_j + out_off[b];
        while (i > 0 || j > 0) {
            const int tt = (i == 0) ? 2 : (j == 0) ? 1 : tr[(size_t)i * (M + 1) + j];
            if (tt == 0) { pi[n] = i - 1; pj[n] = j - 1; i--; j--; }
            else if (tt == 1) { pi[n] = i - 1; pj[n] = -1; i--; }
            else { pi[n] = -1; pj[n] = j - 1; j--; }
            n++;
        }
        for (int x = 0, z = n - 1; x < z; x++, z--) {
            const int ti = pi[x], tj = pj[x]; pi[x] = pi[z]; pj[x] = pj[z]; pi[z] = ti; pj[z] = tj;
        }
        out_len[b] = n;
    }
}

// ---------------------------------------------------------------------------
// Batched Levenshtein distance between words (legacy aligner, Code/Aligners/levenshtein_dist_align_txtgrids.py:43-70):
// previous_row / current_row recurrence, unit costs, min(insertion, deletion, substitution) over Python characters (= Unicode code
// points, here uint32).  Integer arithmetic: the distance is exactly the reference's.  The distance is symmetric, so the shorter
// string takes the rows (the reference's swap at :54-55 has the same effect on its loop nest and none on the value).
//
// One WAVE per pair (four pairs per workgroup): lane = row of a 64-row stripe, the anti-diagonals sweep the stripe, and everything a
// cell needs from its neighbours travels by DPP row shifts -- the cell above came from lane - 1 one step ago, the diagonal one is what
// that shift delivered the step before, the column's character is a shift register fed at lane 0.  No LDS, no barrier.  Words fit one
// stripe; longer strings take ceil(rows / 64) stripes, the last row of a stripe handed to the next through a global row buffer
// (ping-pong, written and read in coalesced 64-element chunks, one device-scope fence per stripe).  No length limit.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_levenshtein(const unsigned *__restrict__ a_chars, const long long *__restrict__ a_off,
                                                     const unsigned *__restrict__ b_chars, const long long *__restrict__ b_off, int batch,
                                                     int *__restrict__ rows, const long long *__restrict__ rows_off, int *__restrict__ out)
{
    const int lane = threadIdx.x & 63, pair = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: scalar loop control)
    if (pair >= batch) return;
    int N = (int)(a_off[pair + 1] - a_off[pair]), M = (int)(b_off[pair + 1] - b_off[pair]);
    const unsigned *A = a_chars + a_off[pair], *B = b_chars + b_off[pair];
    if (N > M) { const unsigned *t = A; A = B; B = t; const int n = N; N = M; M = n; }       // rows = the shorter string
    if (N == 0) { if (lane == 0) out[pair] = M; return; }                                      // (:57-58)
    int *top = rows + rows_off[pair], *bot = top + (M + 1);                                    // only touched when N > 64
    const int n_stripes = (N + 63) >> 6;
    for (int s = 0; s < n_stripes; s++) {
        const int r0 = s << 6, R = min(64, N - r0), last = R - 1;
        const bool more = s + 1 < n_stripes;
        const unsigned my_a = lane < R ? A[r0 + lane] : 0u;
        int cur = r0 + lane + 1;                           // score[r][0] of this lane's row r = r0 + lane + 1
        int up = r0 + lane;                                // score[r - 1][0]
        unsigned bch = 0u;
        // 64-column chunks of the column characters and of the row above the stripe, fetched one chunk ahead
        unsigned chB = lane < M ? B[lane] : 0u, nxB = 0u;
        int chT = s == 0 ? lane + 1 : (lane + 1 <= M ? top[lane + 1] : 0), nxT = 0;
        int botc = r0 + R;                                 // chunk of the stripe's last row being collected; element 0 = score[r0 + R][0]
        const int steps = M + R - 1;
        for (int t = 0; t < steps; t++) {
            const int k = t & 63;
            if (k == 0) {
                if (t) { chB = nxB; chT = nxT; }
                const int c1 = t + 64 + lane;              // next chunk: B[c1], top[c1 + 1]
                nxB = c1 < M ? B[c1] : 0u;
                nxT = s == 0 ? c1 + 1 : (c1 + 1 <= M ? top[c1 + 1] : 0);
            }
            const unsigned b_in = (unsigned)__builtin_amdgcn_readlane((int)chB, k);            // B[t]
            const int top_in = __builtin_amdgcn_readlane(chT, k);                               // score[r0][t + 1]
            const int upn_s = __builtin_amdgcn_update_dpp(0, cur, 0x138, 0xF, 0xF, true);                   // wave_shr:1 = the lane above's value
            const unsigned bch_s = (unsigned)__builtin_amdgcn_update_dpp(0, (int)bch, 0x138, 0xF, 0xF, true);
            const int upn = lane == 0 ? top_in : upn_s;
            bch = lane == 0 ? b_in : bch_s;
            const int c = t - lane + 1;                    // this lane's column at this step
            if (lane < R && c >= 1 && c <= M) {
                const int sub = up + (my_a != bch ? 1 : 0), ins = upn + 1, del = cur + 1;      // (:64-67)
                cur = min(min(ins, del), sub);
                up = upn;
            }
            if (more) {
                const int cb = t - last + 1;               // the column the stripe's last row has just finished
                if (cb >= 1) {
                    const int v = __builtin_amdgcn_readlane(cur, last);
                    if (lane == (cb & 63)) botc = v;
                    if ((cb & 63) == 63 || cb == M) {
                        const int base = cb & ~63;
                        if (lane <= (cb & 63)) bot[base + lane] = botc;
                    }
                }
            }
        }
        if (!more) { if (lane == last) out[pair] = cur; }
        else { __threadfence(); int *x = top; top = bot; bot = x; }
    }
}

} // namespace

// device-resident batch (used by the Whisper alignment path): all pointers are device pointers
int pce_dtw_launch(pce_ctx *c, const double *d_x, int64_t x_stride, int ld, const int *d_rows, const int *d_cols, int N_max, int M_max,
                   int batch, unsigned char *d_trace, int *d_pi, int *d_pj, int *d_pl)
{
    if (N_max > DTW_MAXN) return pce_fail(c, PCE_E_LIMIT, "DTW supports at most %d rows", DTW_MAXN);
    KernelTimer t(c, PCE_K_DTW);
    const int threads = ((N_max + 63) / 64) * 64;
    hipLaunchKernelGGL(k_dtw, dim3((unsigned)batch), dim3((unsigned)threads), 0, c->stream, d_x, x_stride, ld, d_rows, d_cols, N_max, M_max,
                       d_trace, d_pi, d_pj, d_pl);
    PCE_HIP(c, hipGetLastError());
    return PCE_OK;
}

extern "C" {

int pce_dtw(pce_ctx *c, const double *x, int32_t n_rows, int32_t n_cols, int32_t batch, int32_t *path_i, int32_t *path_j, int32_t *path_len)
{
    if (!c || !x || !path_i || !path_j || !path_len || n_rows <= 0 || n_cols <= 0 || batch <= 0) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    const size_t cells = (size_t)batch * n_rows * n_cols, pl = (size_t)batch * (size_t)(n_rows + n_cols);
    DevBuf dx, dtr, dpi, dpj, dpl;
    PCE_HIP(c, dx.reserve(sizeof(double) * cells));
    PCE_HIP(c, dtr.reserve((size_t)batch * (size_t)(n_rows + 1) * (size_t)(n_cols + 1)));
    PCE_HIP(c, dpi.reserve(sizeof(int) * pl)); PCE_HIP(c, dpj.reserve(sizeof(int) * pl)); PCE_HIP(c, dpl.reserve(sizeof(int) * (size_t)batch));
    PCE_HIP(c, hipMemcpyAsync(dx.p, x, sizeof(double) * cells, hipMemcpyHostToDevice, c->stream));
    int rc = pce_dtw_launch(c, dx.as<double>(), (int64_t)n_rows * n_cols, n_cols, nullptr, nullptr, n_rows, n_cols, batch,
                            dtr.as<unsigned char>(), dpi.as<int>(), dpj.as<int>(), dpl.as<int>());
    if (rc) return rc;
    PCE_HIP(c, hipMemcpyAsync(path_i, dpi.p, sizeof(int) * pl, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipMemcpyAsync(path_j, dpj.p, sizeof(int) * pl, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipMemcpyAsync(path_len, dpl.p, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    dx.release(); dtr.release(); dpi.release(); dpj.release(); dpl.release();
    return PCE_OK;
}

int pce_nw_align(pce_ctx *c, const int32_t *a_ids, const int64_t *a_off, const int32_t *b_ids, const int64_t *b_off, int32_t batch,
                 int32_t match, int32_t mismatch, int32_t gap, int32_t *out_i, int32_t *out_j, int32_t *out_len)
{
    if (!c || !a_off || !b_off || !out_i || !out_j || !out_len || batch <= 0) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    std::vector<long long> tro((size_t)batch + 1, 0), oo((size_t)batch + 1, 0), rwo((size_t)batch + 1, 0);
    int max_rows = 0;
    for (int32_t b = 0; b < batch; b++) {
        const int64_t n = a_off[b + 1] - a_off[b], m = b_off[b + 1] - b_off[b];
        if (n < 0 || m < 0) return pce_fail(c, PCE_E_INVALID, "pce_nw_align: offsets of pair %d decrease", b);
        if (n > 0x3fffffff || m > 0x3fffffff) return pce_fail(c, PCE_E_LIMIT, "pce_nw_align: pair %d is longer than 2^30 tokens", b);
        tro[(size_t)b + 1] = tro[(size_t)b] + (n + 1) * (m + 1);
        rwo[(size_t)b + 1] = rwo[(size_t)b] + (n > DTW_MAXN ? 2 * (m + 1) : 0);       // stripe hand-over rows (ping-pong)
        oo[(size_t)b + 1] = oo[(size_t)b] + n + m;
        if (n > max_rows) max_rows = (int)n;
    }
    const size_t na = (size_t)a_off[batch], nb = (size_t)b_off[batch], no = (size_t)oo[(size_t)batch];
    if (a_off[0] != 0 || b_off[0] != 0) return pce_fail(c, PCE_E_INVALID, "pce_nw_align: offsets must start at 0");
    if ((na && !a_ids) || (nb && !b_ids)) return PCE_E_INVALID;
    DevBuf da, db, dao, dbo, dtr, dtro, doi, doj, doo, dol, drw, drwo;
    PCE_HIP(c, drw.reserve(sizeof(int) * ((size_t)rwo[(size_t)batch] + 1))); PCE_HIP(c, drwo.reserve(sizeof(long long) * ((size_t)batch + 1)));
    PCE_HIP(c, da.reserve(sizeof(int) * (na + 1))); PCE_HIP(c, db.reserve(sizeof(int) * (nb + 1)));
    PCE_HIP(c, dao.reserve(sizeof(long long) * ((size_t)batch + 1))); PCE_HIP(c, dbo.reserve(sizeof(long long) * ((size_t)batch + 1)));
    PCE_HIP(c, dtr.reserve((size_t)tro[(size_t)batch] + 1)); PCE_HIP(c, dtro.reserve(sizeof(long long) * ((size_t)batch + 1)));
    PCE_HIP(c, doi.reserve(sizeof(int) * (no + 1))); PCE_HIP(c, doj.reserve(sizeof(int) * (no + 1)));
    PCE_HIP(c, doo.reserve(sizeof(long long) * ((size_t)batch + 1))); PCE_HIP(c, dol.reserve(sizeof(int) * (size_t)batch));
    if (na) PCE_HIP(c, hipMemcpyAsync(da.p, a_ids, sizeof(int) * na, hipMemcpyHostToDevice, c->stream));
    if (nb) PCE_HIP(c, hipMemcpyAsync(db.p, b_ids, sizeof(int) * nb, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dao.p, a_off, sizeof(long long) * ((size_t)batch + 1), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dbo.p, b_off, sizeof(long long) * ((size_t)batch + 1), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dtro.p, tro.data(), sizeof(long long) * tro.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(doo.p, oo.data(), sizeof(long long) * oo.size(), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(drwo.p, rwo.data(), sizeof(long long) * rwo.size(), hipMemcpyHostToDevice, c->stream));
    int threads = 64; while (threads < max_rows && threads < DTW_MAXN) threads <<= 1;
    {
        KernelTimer t(c, PCE_K_NW);
        hipLaunchKernelGGL(k_nw, dim3((unsigned)batch), dim3((unsigned)threads), 0, c->stream, da.as<int>(), dao.as<long long>(), db.as<int>(),
                           dbo.as<long long>(), match, mismatch, gap, dtr.as<unsigned char>(), dtro.as<long long>(), doi.as<int>(), doj.as<int>(),
                           doo.as<long long>(), dol.as<int>(), drw.as<int>(), drwo.as<long long>());
    }
    PCE_HIP(c, hipGetLastError());
    if (no) {
        PCE_HIP(c, hipMemcpyAsync(out_i, doi.p, sizeof(int) * no, hipMemcpyDeviceToHost, c->stream));
        PCE_HIP(c, hipMemcpyAsync(out_j, doj.p, sizeof(int) * no, hipMemcpyDeviceToHost, c->stream));
    }
    PCE_HIP(c, hipMemcpyAsync(out_len, dol.p, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    for (DevBuf *x : {&da, &db, &dao, &dbo, &dtr, &dtro, &doi, &doj, &doo, &dol, &drw, &drwo}) x->release();
    return PCE_OK;
}

int pce_levenshtein(pce_ctx *c, const uint32_t *a_chars, const int64_t *a_off, const uint32_t *b_chars, const int64_t *b_off, int32_t batch,
                    int32_t *out_dist)
{
    if (!c || !a_off || !b_off || !out_dist || batch <= 0) return PCE_E_INVALID;
    PCE_HIP(c, hipSetDevice(c->device));
    if (a_off[0] != 0 || b_off[0] != 0) return pce_fail(c, PCE_E_INVALID, "pce_levenshtein: offsets must start at 0");
    std::vector<long long> ro((size_t)batch + 1, 0);
    for (int32_t b = 0; b < batch; b++) {
        const int64_t n = a_off[b + 1] - a_off[b], m = b_off[b + 1] - b_off[b];
        if (n < 0 || m < 0) return pce_fail(c, PCE_E_INVALID, "pce_levenshtein: offsets of pair %d decrease", b);
        if (n > 0x3fffffff || m > 0x3fffffff) return pce_fail(c, PCE_E_LIMIT, "pce_levenshtein: pair %d is longer than 2^30 characters", b);
        const int64_t shorter = n < m ? n : m, longer = n < m ? m : n;
        ro[(size_t)b + 1] = ro[(size_t)b] + (shorter > 64 ? 2 * (longer + 1) : 0);          // stripe hand-over rows (ping-pong)
    }
    const size_t na = (size_t)a_off[batch], nb = (size_t)b_off[batch];
    if ((na && !a_chars) || (nb && !b_chars)) return PCE_E_INVALID;
    DevBuf da, db, dao, dbo, dro, drows, dout;
    PCE_HIP(c, da.reserve(sizeof(unsigned) * (na + 1))); PCE_HIP(c, db.reserve(sizeof(unsigned) * (nb + 1)));
    PCE_HIP(c, dao.reserve(sizeof(long long) * ((size_t)batch + 1))); PCE_HIP(c, dbo.reserve(sizeof(long long) * ((size_t)batch + 1)));
    PCE_HIP(c, dro.reserve(sizeof(long long) * ((size_t)batch + 1))); PCE_HIP(c, drows.reserve(sizeof(int) * ((size_t)ro[(size_t)batch] + 1)));
    PCE_HIP(c, dout.reserve(sizeof(int) * (size_t)batch));
    if (na) PCE_HIP(c, hipMemcpyAsync(da.p, a_chars, sizeof(unsigned) * na, hipMemcpyHostToDevice, c->stream));
    if (nb) PCE_HIP(c, hipMemcpyAsync(db.p, b_chars, sizeof(unsigned) * nb, hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dao.p, a_off, sizeof(long long) * ((size_t)batch + 1), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dbo.p, b_off, sizeof(long long) * ((size_t)batch + 1), hipMemcpyHostToDevice, c->stream));
    PCE_HIP(c, hipMemcpyAsync(dro.p, ro.data(), sizeof(long long) * ro.size(), hipMemcpyHostToDevice, c->stream));
    {
        KernelTimer t(c, PCE_K_LEVENSHTEIN);
        hipLaunchKernelGGL(k_levenshtein, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, c->stream, da.as<unsigned>(), dao.as<long long>(),
                           db.as<unsigned>(), dbo.as<long long>(), batch, drows.as<int>(), dro.as<long long>(), dout.as<int>());
    }
    PCE_HIP(c, hipGetLastError());
    PCE_HIP(c, hipMemcpyAsync(out_dist, dout.p, sizeof(int) * (size_t)batch, hipMemcpyDeviceToHost, c->stream));
    PCE_HIP(c, hipStreamSynchronize(c->stream));
    pce_profile_collect(c);
    for (DevBuf *x : {&da, &db, &dao, &dbo, &dro, &drows, &dout}) x->release();
    return PCE_OK;
}

} // extern "C"
