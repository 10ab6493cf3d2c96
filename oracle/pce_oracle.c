/*
 * oracle/pce_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU (double precision, scalar, single thread) restatement of the numerics
 * on the hot path of hi-paris/Prosody-Control-French-TTS.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY STATUS: "parity unpinned" for the third-party numerics.  The
 * reference delegates this arithmetic to packages whose source is not under
 * /root/reference and which are not installed here:
 *
 *   praat-parselmouth==0.4.5  (tts-env.yml:166)   Sound.to_pitch / extract_part
 *        call sites: Code/audioPipeline.py:326-335,
 *                    Code/Pipeline/compute_pitch_adjustments.py:167-208
 *   pyloudnorm (unpinned)     Meter.integrated_loudness
 *        call sites: Code/audioPipeline.py:338-358, :373, :493
 *
 * What is restated is the published algorithm of those packages:
 *   - Praat "Sound: To Pitch (ac)..." (Boersma 1993; fon/Sound_to_Pitch.cpp:
 *     Sound_to_Pitch_any with method AC_HANNING, NUM_interpolate_sinc,
 *     NUMimproveExtremum + Brent minimiser, Pitch_pathFinder), with the
 *     parameter set parselmouth's to_pitch(pitch_floor, pitch_ceiling) maps to
 *     (time step 0 -> 0.75/floor, 3 periods/window, 15 candidates, silence
 *     threshold 0.03, voicing threshold 0.45, octave cost 0.01, octave-jump
 *     cost 0.35, voiced/unvoiced cost 0.14).
 *   - ITU-R BS.1770-4 integrated loudness as pyloudnorm implements it
 *     (RBJ-cookbook high-shelf + high-pass biquads designed per sample rate,
 *     direct-form-II-transposed filtering as scipy.signal.lfilter, 400 ms
 *     blocks / 75 % overlap, -70 LKFS absolute and -10 LU relative gates).
 * They are pinned by analytic known-answer tests (tests/test_oracle_*.py) and,
 * for the LUFS filter, against scipy.signal.lfilter which IS installed.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no fast-math)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <float.h>

#define PI 3.1415926535897932384626433832795028841972
#define NUMlog2e 1.4426950408889634073599246810018921374266
#define GOLDEN_SECTION 0.6180339887498948482045868343656381177203

/* ------------------------------------------------------------------ */
/* complex radix-2 FFT, double (stands in for Praat's FFTPACK real FFT) */
/* ------------------------------------------------------------------ */
typedef struct { long n; double *cosv, *sinv; long *rev; } fft_table;

static void fft_init(fft_table *t, long n)
{
    long bits = 0; while ((1L << bits) < n) bits++;
    t->n = n;
    t->cosv = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
    t->sinv = (double *)malloc(sizeof(double) * (size_t)(n / 2 + 1));
    t->rev  = (long *)malloc(sizeof(long) * (size_t)n);
    for (long k = 0; k < n / 2; k++) {
        t->cosv[k] = cos(2.0 * PI * (double)k / (double)n);
        t->sinv[k] = sin(2.0 * PI * (double)k / (double)n);
    }
    for (long i = 0; i < n; i++) {
        long r = 0;
        for (long b = 0; b < bits; b++) if (i & (1L << b)) r |= 1L << (bits - 1 - b);
        t->rev[i] = r;
    }
}
static void fft_free(fft_table *t) { free(t->cosv); free(t->sinv); free(t->rev); }

/* in-place forward transform (sign = -1) or unnormalised inverse (sign = +1) */
static void fft_run(const fft_table *t, double *re, double *im, int sign)
{
    long n = t->n;
    for (long i = 0; i < n; i++) {
        long j = t->rev[i];
        if (j > i) { double a = re[i]; re[i] = re[j]; re[j] = a; a = im[i]; im[i] = im[j]; im[j] = a; }
    }
    for (long len = 2; len <= n; len <<= 1) {
        long half = len >> 1, step = n / len;
        for (long i = 0; i < n; i += len)
            for (long k = 0; k < half; k++) {
                double wr = t->cosv[k * step], wi = (double)sign * t->sinv[k * step];
                double xr = re[i + k + half], xi = im[i + k + half];
                double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
                re[i + k + half] = re[i + k] - tr; im[i + k + half] = im[i + k] - ti;
                re[i + k] += tr; im[i + k] += ti;
            }
    }
}

/* autocorrelation of a zero-padded real frame: ac[0..n-1] (scaled by n, as
 * Praat's unnormalised forward+backward FFT pair is; the scale cancels). */
static void autocorr_fft(const fft_table *t, const double *frame, double *ac, double *wre, double *wim)
{
    long n = t->n;
    for (long i = 0; i < n; i++) { wre[i] = frame[i]; wim[i] = 0.0; }
    fft_run(t, wre, wim, -1);
    for (long i = 0; i < n; i++) { wre[i] = wre[i] * wre[i] + wim[i] * wim[i]; wim[i] = 0.0; }
    fft_run(t, wre, wim, +1);
    for (long i = 0; i < n; i++) ac[i] = wre[i];
}

/* ------------------------------------------------------------------ */
/* Praat NUM_interpolate_sinc (melder/NUMinterpol.cpp); y is 1-based   */
/* ------------------------------------------------------------------ */
static double interpolate_sinc(const double *y /*1-based*/, long nx, double x, long maxDepth)
{
    long ix, midleft = (long)floor(x), midright = midleft + 1, left, right;
    double result = 0.0, a, halfsina, aa, daa;
    if (nx < 1) return NAN;
    if (x > (double)nx) return y[nx];
    if (x < 1.0) return y[1];
    if (x == (double)midleft) return y[midleft];
    if (maxDepth > midright - 1) maxDepth = midright - 1;
    if (maxDepth > nx - midleft) maxDepth = nx - midleft;
    if (maxDepth <= 0) return y[(long)floor(x + 0.5)];
    if (maxDepth == 1) return y[midleft] + (x - (double)midleft) * (y[midright] - y[midleft]);
    if (maxDepth == 2) {
        double yl = y[midleft], yr = y[midright];
        double dyl = 0.5 * (yr - y[midleft - 1]), dyr = 0.5 * (y[midright + 1] - yl);
        double fil = x - (double)midleft, fir = (double)midright - x;
        return yl * fir + yr * fil - fil * fir * (0.5 * (dyr - dyl) + (fil - 0.5) * (dyl + dyr - 2.0 * (yr - yl)));
    }
    left = midright - maxDepth; right = midleft + maxDepth;
    a = PI * (x - (double)midleft);
    halfsina = 0.5 * sin(a);
    aa = a / (x - (double)left + 1.0);
    daa = PI / (x - (double)left + 1.0);
    for (ix = midleft; ix >= left; ix--) {
        double d = halfsina / a * (1.0 + cos(aa));
        result += y[ix] * d;
        a += PI; aa += daa; halfsina = -halfsina;
    }
    a = PI * ((double)midright - x);
    halfsina = 0.5 * sin(a);
    aa = a / ((double)right - x + 1.0);
    daa = PI / ((double)right - x + 1.0);
    for (ix = midright; ix <= right; ix++) {
        double d = halfsina / a * (1.0 + cos(aa));
        result += y[ix] * d;
        a += PI; aa += daa; halfsina = -halfsina;
    }
    return result;
}

/* ------------------------------------------------------------------ */
/* Praat NUMimproveExtremum / NUMminimize_brent, maximum case          */
/* ------------------------------------------------------------------ */
typedef struct { const double *y; long nx; long depth; long evals; } improve_ctx;
static double improve_eval(double x, improve_ctx *c) { c->evals++; return -interpolate_sinc(c->y, c->nx, x, c->depth); }

static double minimize_brent(improve_ctx *c, double a, double b, double tol, double *fx)
{
    double x, v, fv, w, fw;
    const double golden = 1.0 - GOLDEN_SECTION;
    const double sqrt_epsilon = sqrt(DBL_EPSILON);
    const long itermax = 60;
    v = a + golden * (b - a);
    fv = improve_eval(v, c);
    x = v; w = v; *fx = fv; fw = fv;
    for (long iter = 1; iter <= itermax; iter++) {
        double range = b - a;
        double middle_range = (a + b) / 2.0;
        double tol_act = sqrt_epsilon * fabs(x) + tol / 3.0;
        double new_step;
        if (fabs(x - middle_range) + range / 2.0 <= 2.0 * tol_act) return x;
        new_step = golden * (x < middle_range ? b - x : a - x);
        if (fabs(x - w) >= tol_act) {
            double p, q, t;
            t = (x - w) * (*fx - fv);
            q = (x - v) * (*fx - fw);
            p = (x - v) * q - (x - w) * t;
            q = 2.0 * (q - t);
            if (q > 0.0) p = -p; else q = -q;
            if (fabs(p) < fabs(new_step * q) && p > q * (a - x + 2.0 * tol_act) && p < q * (b - x - 2.0 * tol_act))
                new_step = p / q;
        }
        if (fabs(new_step) < tol_act) new_step = new_step > 0.0 ? tol_act : -tol_act;
        {
            double t = x + new_step;
            double ft = improve_eval(t, c);
            if (ft <= *fx) {
                if (t < x) b = x; else a = x;
                v = w; w = x; x = t;
                fv = fw; fw = *fx; *fx = ft;
            } else {
                if (t < x) a = t; else b = t;
                if (ft <= fw || w == x) { v = w; w = t; fv = fw; fw = ft; }
                else if (ft <= fv || v == x || v == w) { v = t; fv = ft; }
            }
        }
    }
    return x;
}

/* interpolation: 3 = SINC70, 4 = SINC700 */
static double improve_maximum(const double *y /*1-based*/, long nx, long ixmid, int interpolation, double *ixmid_real, long *evals)
{
    improve_ctx c; double result;
    if (ixmid <= 1) { *ixmid_real = 1.0; return y[1]; }
    if (ixmid >= nx) { *ixmid_real = (double)nx; return y[nx]; }
    c.y = y; c.nx = nx; c.depth = interpolation == 3 ? 70 : 700; c.evals = 0;
    *ixmid_real = minimize_brent(&c, (double)(ixmid - 1), (double)(ixmid + 1), 1e-10, &result);
    if (evals) *evals += c.evals;
    return -result;
}

/* ------------------------------------------------------------------ */
/* Sound_to_Pitch_ac                                                   */
/* ------------------------------------------------------------------ */
typedef struct {
    double time_step;          /* <= 0: 0.75 / pitch_floor                     */
    double pitch_floor;        /* minimumPitch                                  */
    double periods_per_window; /* 3.0                                           */
    int32_t max_candidates;    /* 15                                            */
    int32_t reserved;
    double silence_threshold;  /* 0.03 */
    double voicing_threshold;  /* 0.45 */
    double octave_cost;        /* 0.01 */
    double octave_jump_cost;   /* 0.35 */
    double voiced_unvoiced_cost; /* 0.14 */
    double pitch_ceiling;      /* 600 */
} por_pitch_params;

typedef struct {
    double dt, t1, ceiling, dt_window;
    long n_frames, nsamp_period, halfnsamp_period, nsamp_window, halfnsamp_window;
    long maximum_lag, nsamp_fft, brent_ixmax, max_candidates;
} por_pitch_plan;

enum { POR_OK = 0, POR_E_TOO_SHORT = -1, POR_E_WINDOW = -2, POR_E_ARG = -3 };

/* Derives every size exactly as Sound_to_Pitch_any does before its frame loop.
 * nx samples, sampling period dx, first sample time x1. */
int por_pitch_plan_make(long nx, double dx, double x1, const por_pitch_params *p, por_pitch_plan *pl)
{
    double dt = p->time_step, minimumPitch = p->pitch_floor, periodsPerWindow = p->periods_per_window;
    double ceiling = p->pitch_ceiling, duration, myDuration, ourMidTime, thyDuration;
    long maxnCandidates = p->max_candidates;
    if (nx < 1 || !(dx > 0.0) || !(minimumPitch > 0.0)) return POR_E_ARG;
    if (maxnCandidates < 2) maxnCandidates = 2;
    if ((double)maxnCandidates < ceiling / minimumPitch) maxnCandidates = (long)floor(ceiling / minimumPitch);
    if (dt <= 0.0) dt = periodsPerWindow / minimumPitch / 4.0;
    duration = dx * (double)nx;
    if (minimumPitch < periodsPerWindow / duration) return POR_E_TOO_SHORT;
    pl->nsamp_period = (long)floor(1.0 / dx / minimumPitch);
    pl->halfnsamp_period = pl->nsamp_period / 2 + 1;
    if (ceiling > 0.5 / dx) ceiling = 0.5 / dx;
    pl->dt_window = periodsPerWindow / minimumPitch;
    pl->nsamp_window = (long)floor(pl->dt_window / dx);
    pl->halfnsamp_window = pl->nsamp_window / 2 - 1;
    if (pl->halfnsamp_window < 2) return POR_E_WINDOW;
    pl->nsamp_window = pl->halfnsamp_window * 2;
    pl->maximum_lag = (long)floor((double)pl->nsamp_window / periodsPerWindow) + 2;
    if (pl->maximum_lag > pl->nsamp_window) pl->maximum_lag = pl->nsamp_window;
    /* Sampled_shortTermAnalysis (me, dt_window, dt, &nFrames, &t1) */
    myDuration = dx * (double)nx;
    if (pl->dt_window > myDuration) return POR_E_TOO_SHORT;
    pl->n_frames = (long)floor((myDuration - pl->dt_window) / dt) + 1;
    if (pl->n_frames < 1) return POR_E_TOO_SHORT;
    ourMidTime = x1 - 0.5 * dx + 0.5 * myDuration;
    thyDuration = (double)pl->n_frames * dt;
    pl->t1 = ourMidTime - 0.5 * thyDuration + 0.5 * dt;
    pl->dt = dt; pl->ceiling = ceiling; pl->max_candidates = maxnCandidates;
    pl->nsamp_fft = 1;
    while ((double)pl->nsamp_fft < (double)pl->nsamp_window * (1.0 + 0.5)) pl->nsamp_fft *= 2;
    pl->brent_ixmax = (long)((double)pl->nsamp_window * 0.5);
    return POR_OK;
}

/* Hanning window and its normalised autocorrelation windowR[0..brent_ixmax]. */
static void make_window(const por_pitch_plan *pl, const fft_table *ft, double *window /*[nsamp_window]*/, double *windowR /*[nsamp_fft]*/)
{
    long nw = pl->nsamp_window, nf = pl->nsamp_fft;
    double *tmp = (double *)calloc((size_t)nf * 3, sizeof(double));
    for (long i = 1; i <= nw; i++) window[i - 1] = 0.5 - 0.5 * cos((double)i * 2.0 * PI / (double)(nw + 1));
    for (long i = 0; i < nw; i++) tmp[i] = window[i];
    autocorr_fft(ft, tmp, windowR, tmp + nf, tmp + 2 * nf);
    for (long i = 1; i < nw; i++) windowR[i] /= windowR[0];
    windowR[0] = 1.0;
    free(tmp);
}

/* exported so the host side of the product can be checked against it */
void por_window_autocorr(long nx, double dx, double x1, const por_pitch_params *p, double *windowR_out, long n_out)
{
    por_pitch_plan pl; fft_table ft;
    if (por_pitch_plan_make(nx, dx, x1, p, &pl) != POR_OK) return;
    fft_init(&ft, pl.nsamp_fft);
    double *window = (double *)malloc(sizeof(double) * (size_t)pl.nsamp_window);
    double *windowR = (double *)malloc(sizeof(double) * (size_t)pl.nsamp_fft);
    make_window(&pl, &ft, window, windowR);
    for (long i = 0; i < n_out && i < pl.nsamp_fft; i++) windowR_out[i] = windowR[i];
    free(window); free(windowR); fft_free(&ft);
}

/*
 * z[0..nx-1]: samples (already scaled to [-1,1)), mono.
 * Outputs (caller-allocated, n_frames from por_pitch_plan_make):
 *   f0[n_frames]        selected candidate frequency after path finding (0 = unvoiced)
 *   strength[n_frames]  its strength
 *   intensity[n_frames] frame intensity (localPeak/globalPeak, <= 1)
 *   cand_f / cand_s     [n_frames * max_candidates] candidates BEFORE path finding (may be NULL)
 *   ncand[n_frames]     (may be NULL)
 *   stats[0] = total Brent evaluations, stats[1] = total candidates (may be NULL)
 */
int por_pitch_ac(const double *z, long nx, double dx, double x1, const por_pitch_params *p,
                 double *f0, double *strength, double *intensity,
                 double *cand_f_out, double *cand_s_out, int32_t *ncand_out, int64_t *stats)
{
    por_pitch_plan pl; fft_table ft;
    int st = por_pitch_plan_make(nx, dx, x1, p, &pl);
    if (st != POR_OK) return st;
    const long nF = pl.n_frames, maxc = pl.max_candidates, nw = pl.nsamp_window, nf = pl.nsamp_fft, bix = pl.brent_ixmax;
    const double minimumPitch = p->pitch_floor, voicingThreshold = p->voicing_threshold, octaveCost = p->octave_cost;
    const double ceiling = pl.ceiling;
    long total_evals = 0, total_cands = 0;

    double *cf = (double *)calloc((size_t)(nF * maxc), sizeof(double));
    double *cs = (double *)calloc((size_t)(nF * maxc), sizeof(double));
    int32_t *nc = (int32_t *)malloc(sizeof(int32_t) * (size_t)nF);
    for (long i = 0; i < nF; i++) { nc[i] = 1; intensity[i] = 0.0; }

    /* global absolute peak around the global mean */
    double sum = 0.0, globalPeak = 0.0;
    for (long i = 0; i < nx; i++) sum += z[i];
    double mean = sum / (double)nx;
    for (long i = 0; i < nx; i++) { double v = fabs(z[i] - mean); if (v > globalPeak) globalPeak = v; }

    if (globalPeak != 0.0) {
        fft_init(&ft, nf);
        double *window = (double *)malloc(sizeof(double) * (size_t)nw);
        double *windowR = (double *)malloc(sizeof(double) * (size_t)nf);
        double *frame = (double *)malloc(sizeof(double) * (size_t)nf);
        double *ac = (double *)malloc(sizeof(double) * (size_t)nf);
        double *wre = (double *)malloc(sizeof(double) * (size_t)nf);
        double *wim = (double *)malloc(sizeof(double) * (size_t)nf);
        double *rbuf = (double *)malloc(sizeof(double) * (size_t)(2 * bix + 1));
        double *r = rbuf + bix;                 /* r[-bix..bix] */
        long *imax = (long *)malloc(sizeof(long) * (size_t)(maxc + 1));
        make_window(&pl, &ft, window, windowR);

        for (long iframe = 1; iframe <= nF; iframe++) {
            double *fcf = cf + (iframe - 1) * maxc - 1, *fcs = cs + (iframe - 1) * maxc - 1; /* 1-based */
            double t = pl.t1 + (double)(iframe - 1) * pl.dt;            /* Sampled_indexToX */
            long leftSample = (long)floor((t - x1) / dx) + 1;          /* Sampled_xToLowIndex, 1-based */
            long rightSample = leftSample + 1;
            long startSample, endSample;
            double localMean = 0.0, localPeak = 0.0;
            int n = 1;

            startSample = rightSample - pl.nsamp_period;
            endSample = leftSample + pl.nsamp_period;
            if (startSample < 1 || endSample > nx) { st = POR_E_ARG; goto done; }   /* Melder_assert */
            for (long i = startSample; i <= endSample; i++) localMean += z[i - 1];
            localMean /= (double)(2 * pl.nsamp_period);

            startSample = rightSample - pl.halfnsamp_window;
            endSample = leftSample + pl.halfnsamp_window;
            if (startSample < 1 || endSample > nx) { st = POR_E_ARG; goto done; }
            for (long j = 1, i = startSample; j <= nw; j++) frame[j - 1] = (z[i++ - 1] - localMean) * window[j - 1];
            for (long j = nw + 1; j <= nf; j++) frame[j - 1] = 0.0;

            if ((startSample = pl.halfnsamp_window + 1 - pl.halfnsamp_period) < 1) startSample = 1;
            if ((endSample = pl.halfnsamp_window + pl.halfnsamp_period) > nw) endSample = nw;
            for (long j = startSample; j <= endSample; j++) { double v = fabs(frame[j - 1]); if (v > localPeak) localPeak = v; }
            intensity[iframe - 1] = localPeak > globalPeak ? 1.0 : localPeak / globalPeak;

            fcf[1] = 0.0; fcs[1] = 0.0;
            if (localPeak == 0.0) { nc[iframe - 1] = 1; continue; }

            autocorr_fft(&ft, frame, ac, wre, wim);
            r[0] = 1.0;
            for (long i = 1; i <= bix; i++) r[-i] = r[i] = ac[i] / (ac[0] * windowR[i]);

            imax[1] = 0;
            for (long i = 2; i < pl.maximum_lag && i < bix; i++)
                if (r[i] > 0.5 * voicingThreshold && r[i] > r[i - 1] && r[i] >= r[i + 1]) {
                    int place = 0;
                    double dr = 0.5 * (r[i + 1] - r[i - 1]), d2r = 2.0 * r[i] - r[i - 1] - r[i + 1];
                    double frequencyOfMaximum = 1.0 / dx / ((double)i + dr / d2r);
                    long offset = -bix - 1;
                    double strengthOfMaximum = interpolate_sinc(&r[offset], bix - offset,
                                                                1.0 / dx / frequencyOfMaximum - (double)offset, 30);
                    if (strengthOfMaximum > 1.0) strengthOfMaximum = 1.0 / strengthOfMaximum;
                    if (n < maxc) {
                        place = ++n;
                    } else {
                        double weakest = 2.0;
                        for (int iweak = 2; iweak <= maxc; iweak++) {
                            double localStrength = fcs[iweak] - octaveCost * (log(minimumPitch / fcf[iweak]) * NUMlog2e);
                            if (localStrength < weakest) { weakest = localStrength; place = iweak; }
                        }
                        if (strengthOfMaximum - octaveCost * (log(minimumPitch / frequencyOfMaximum) * NUMlog2e) <= weakest) place = 0;
                    }
                    if (place) { fcf[place] = frequencyOfMaximum; fcs[place] = strengthOfMaximum; imax[place] = i; }
                }

            for (int i = 2; i <= n; i++) {
                double xmid, ymid;
                long offset = -bix - 1;
                ymid = improve_maximum(&r[offset], bix - offset, imax[i] - offset,
                                       fcf[i] > 0.3 / dx ? 4 : 3, &xmid, &total_evals);
                xmid += (double)offset;
                fcf[i] = 1.0 / dx / xmid;
                if (ymid > 1.0) ymid = 1.0 / ymid;
                fcs[i] = ymid;
            }
            nc[iframe - 1] = n;
            total_cands += n - 1;
        }
done:
        free(window); free(windowR); free(frame); free(ac); free(wre); free(wim); free(rbuf); free(imax);
        fft_free(&ft);
        if (st != POR_OK) { free(cf); free(cs); free(nc); return st; }
    }

    if (cand_f_out) memcpy(cand_f_out, cf, sizeof(double) * (size_t)(nF * maxc));
    if (cand_s_out) memcpy(cand_s_out, cs, sizeof(double) * (size_t)(nF * maxc));
    if (ncand_out) memcpy(ncand_out, nc, sizeof(int32_t) * (size_t)nF);
    if (stats) { stats[0] = total_evals; stats[1] = total_cands; }

    /* ---- Pitch_pathFinder (fon/Pitch.cpp), pullFormants = false ---- */
    {
        double silenceThreshold = p->silence_threshold;
        double octaveJumpCost = p->octave_jump_cost, voicedUnvoicedCost = p->voiced_unvoiced_cost;
        double ceiling2 = ceiling;
        double timeStepCorrection = 0.01 / pl.dt;
        octaveJumpCost *= timeStepCorrection;
        voicedUnvoicedCost *= timeStepCorrection;
        double *delta = (double *)malloc(sizeof(double) * (size_t)(nF * maxc));
        int32_t *psi = (int32_t *)calloc((size_t)(nF * maxc), sizeof(int32_t));
        for (long iframe = 0; iframe < nF; iframe++) {
            double unvoicedStrength = silenceThreshold <= 0.0 ? 0.0 :
                2.0 - intensity[iframe] / (silenceThreshold / (1.0 + voicingThreshold));
            unvoicedStrength = voicingThreshold + (unvoicedStrength > 0.0 ? unvoicedStrength : 0.0);
            for (int ic = 0; ic < nc[iframe]; ic++) {
                double f = cf[iframe * maxc + ic];
                int voiceless = f == 0.0 || f > ceiling2;
                delta[iframe * maxc + ic] = voiceless ? unvoicedStrength :
                    cs[iframe * maxc + ic] - octaveCost * (log(ceiling / f) * NUMlog2e);
            }
        }
        for (long iframe = 1; iframe < nF; iframe++) {
            double *prevDelta = delta + (iframe - 1) * maxc, *curDelta = delta + iframe * maxc;
            for (int ic2 = 0; ic2 < nc[iframe]; ic2++) {
                double f2 = cf[iframe * maxc + ic2];
                double maximum = -1e30; int place = 0;
                for (int ic1 = 0; ic1 < nc[iframe - 1]; ic1++) {
                    double f1 = cf[(iframe - 1) * maxc + ic1], transitionCost, value;
                    int previousVoiceless = f1 <= 0.0 || f1 >= ceiling2;
                    int currentVoiceless = f2 <= 0.0 || f2 >= ceiling2;
                    if (currentVoiceless) transitionCost = previousVoiceless ? 0.0 : voicedUnvoicedCost;
                    else if (previousVoiceless) transitionCost = voicedUnvoicedCost;
                    else transitionCost = octaveJumpCost * fabs(log(f1 / f2) * NUMlog2e);
                    value = prevDelta[ic1] - transitionCost + curDelta[ic2];
                    if (value > maximum) { maximum = value; place = ic1; }
                }
                curDelta[ic2] = maximum;
                psi[iframe * maxc + ic2] = place;
            }
        }
        int place = 0;
        double maximum = delta[(nF - 1) * maxc];
        for (int ic = 1; ic < nc[nF - 1]; ic++)
            if (delta[(nF - 1) * maxc + ic] > maximum) { place = ic; maximum = delta[(nF - 1) * maxc + ic]; }
        for (long iframe = nF - 1; iframe >= 0; iframe--) {
            f0[iframe] = cf[iframe * maxc + place];
            strength[iframe] = cs[iframe * maxc + place];
            place = psi[iframe * maxc + place];
        }
        free(delta); free(psi);
    }
    free(cf); free(cs); free(nc);
    return POR_OK;
}

/* ------------------------------------------------------------------ */
/* BS.1770 integrated loudness, pyloudnorm semantics                   */
/* ------------------------------------------------------------------ */
/* numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src) of x[i]*x[i] */
static double pairwise_sumsq(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i] * a[i];
        return res;
    } else if (n <= 128) {
        double r[8]; long i;
        for (int k = 0; k < 8; k++) r[k] = a[k] * a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k] * a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i] * a[i];
        return res;
    } else {
        long n2 = n / 2; n2 -= n2 % 8;
        return pairwise_sumsq(a, n2) + pairwise_sumsq(a + n2, n - n2);
    }
}

/* numpy pairwise summation of plain values (np.mean of the gated block list) */
static double pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8]; long i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2; n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* K-weighting biquad design exactly as pyloudnorm.IIRfilter.generate_coefficients */
void por_kweight_coeffs(double rate, double *b1 /*3*/, double *a1 /*3*/, double *b2, double *a2)
{
    {   /* high_shelf: G=4.0, Q=1/sqrt(2), fc=1500 */
        double G = 4.0, Q = 1.0 / sqrt(2.0), fc = 1500.0;
        double A = pow(10.0, G / 40.0), w0 = 2.0 * PI * (fc / rate), alpha = sin(w0) / (2.0 * Q);
        double b0 = A * ((A + 1) + (A - 1) * cos(w0) + 2 * sqrt(A) * alpha);
        double bb1 = -2 * A * ((A - 1) + (A + 1) * cos(w0));
        double bb2 = A * ((A + 1) + (A - 1) * cos(w0) - 2 * sqrt(A) * alpha);
        double a0 = (A + 1) - (A - 1) * cos(w0) + 2 * sqrt(A) * alpha;
        double aa1 = 2 * ((A - 1) - (A + 1) * cos(w0));
        double aa2 = (A + 1) - (A - 1) * cos(w0) - 2 * sqrt(A) * alpha;
        b1[0] = b0 / a0; b1[1] = bb1 / a0; b1[2] = bb2 / a0;
        a1[0] = a0 / a0; a1[1] = aa1 / a0; a1[2] = aa2 / a0;
    }
    {   /* high_pass: G=0, Q=0.5, fc=38 */
        double Q = 0.5, fc = 38.0;
        double w0 = 2.0 * PI * (fc / rate), alpha = sin(w0) / (2.0 * Q);
        double b0 = (1 + cos(w0)) / 2, bb1 = -(1 + cos(w0)), bb2 = (1 + cos(w0)) / 2;
        double a0 = 1 + alpha, aa1 = -2 * cos(w0), aa2 = 1 - alpha;
        b2[0] = b0 / a0; b2[1] = bb1 / a0; b2[2] = bb2 / a0;
        a2[0] = a0 / a0; a2[1] = aa1 / a0; a2[2] = aa2 / a0;
    }
}

/* direct form II transposed, as scipy.signal.lfilter */
static void lfilter_biquad(const double *b, const double *a, double *x, long n)
{
    double z0 = 0.0, z1 = 0.0;
    for (long i = 0; i < n; i++) {
        double xi = x[i];
        double y = b[0] * xi + z0;
        z0 = b[1] * xi - a[1] * y + z1;
        z1 = b[2] * xi - a[2] * y;
        x[i] = y;
    }
}

/* Number of gating blocks and their [l,u) bounds exactly as pyloudnorm computes them. */
long por_lufs_num_blocks(long n, double rate)
{
    double T_g = 0.400, overlap = 0.75, step = 1.0 - overlap;
    double T = (double)n / rate;
    return (long)rint((T - T_g) / (T_g * step)) + 1;   /* np.round = half-to-even = rint */
}
void por_lufs_block_bounds(long j, double rate, long *l, long *u)
{
    double T_g = 0.400, overlap = 0.75, step = 1.0 - overlap;
    *l = (long)(T_g * ((double)j * step) * rate);
    *u = (long)(T_g * ((double)j * step + 1.0) * rate);
}

/*
 * samples[0..n-1] are the caller's slice BEFORE peak normalisation (float64
 * sample values as pydub's get_array_of_samples gives them, i.e. int16 range).
 * Mirrors Code/audioPipeline.py:349-352: x/peak (peak = max|x| or 1.0) then
 * Meter(rate).integrated_loudness.  Returns POR_E_TOO_SHORT where pyloudnorm
 * raises ValueError (n < 0.4*rate).  *lufs may be -inf.
 * z_out (may be NULL): per-block mean squares [num_blocks].
 */
int por_lufs(const double *samples, long n, double rate, double *lufs, double *z_out)
{
    if ((double)n < 0.400 * rate) return POR_E_TOO_SHORT;
    double peak = 0.0;
    for (long i = 0; i < n; i++) { double v = fabs(samples[i]); if (v > peak) peak = v; }
    if (peak == 0.0) peak = 1.0;
    double *x = (double *)malloc(sizeof(double) * (size_t)n);
    for (long i = 0; i < n; i++) x[i] = samples[i] / peak;
    double b1[3], a1[3], b2[3], a2[3];
    por_kweight_coeffs(rate, b1, a1, b2, a2);
    lfilter_biquad(b1, a1, x, n);
    lfilter_biquad(b2, a2, x, n);

    long nb = por_lufs_num_blocks(n, rate);
    if (nb < 0) nb = 0;
    double *z = (double *)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
    double T_g = 0.400;
    for (long j = 0; j < nb; j++) {
        long l, u; por_lufs_block_bounds(j, rate, &l, &u);
        if (l > n) l = n;
        if (u > n) u = n;                                   /* numpy slice clipping */
        z[j] = (1.0 / (T_g * rate)) * pairwise_sumsq(x + l, u > l ? u - l : 0);
        if (z_out) z_out[j] = z[j];
    }
    double Gamma_a = -70.0;
    /* first gate: absolute */
    double *g = (double *)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
    long cnt = 0;
    for (long j = 0; j < nb; j++) { double lj = -0.691 + 10.0 * log10(z[j]); if (lj >= Gamma_a) g[cnt++] = z[j]; }
    double z_avg = cnt ? pairwise_sum(g, cnt) / (double)cnt : NAN;   /* np.mean([]) = nan */
    double Gamma_r = -0.691 + 10.0 * log10(z_avg) - 10.0;
    cnt = 0;
    for (long j = 0; j < nb; j++) { double lj = -0.691 + 10.0 * log10(z[j]); if (lj > Gamma_r && lj > Gamma_a) g[cnt++] = z[j]; }
    z_avg = cnt ? pairwise_sum(g, cnt) / (double)cnt : 0.0;          /* nan_to_num(nan) = 0 */
    free(g);
    if (isnan(z_avg)) z_avg = 0.0;
    *lufs = -0.691 + 10.0 * log10(z_avg);
    free(x); free(z);
    return POR_OK;
}
