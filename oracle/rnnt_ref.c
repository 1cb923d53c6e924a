/*
 * oracle/rnnt_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's transducer loss for the hot path
 * (SURVEY.md §8 a14).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object; the product path
 * (indic_cl_asr_amd/) never does.
 *
 * Follows, function by function:
 *   log-softmax front end ........ K/rnnt_pytorch.py:411-437 (explicit log_softmax on CPU),
 *                                  K/utils/cuda_utils/reduce.py:121-248 (denom = -max - log sum exp(x-max))
 *   log_sum_exp .................. K/utils/rnnt_helper.py:42-53
 *   alphas ....................... K/utils/cpu_utils/cpu_rnnt.py:246-276, K/rnnt_numpy.py:112-140
 *   betas ........................ cpu_rnnt.py:278-329, rnnt_numpy.py:143-172
 *   grads w.r.t. log-probs ....... cpu_rnnt.py:331-345, rnnt_numpy.py:175-207
 *   grads w.r.t. logits .......... K/utils/cuda_utils/gpu_rnnt_kernel.py:351-403 (fused log-softmax grad,
 *                                  FastEmit term, clamp)
 *   cost = -ll*(1+lambda) ........ cpu_rnnt.py:236-244, rnnt_helper.py:106-116
 * (K/ = NeMo/nemo/collections/asr/parts/numba/rnnt_loss/)
 *
 * Pinned against the reference's inline known answers and against rnnt_numpy.py
 * outputs generated in the build container (tests/golden/, tests/test_oracle_rnnt.py).
 * All arithmetic is float32 like the reference's kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float lse2(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    if (a > b) return log1pf(expf(b - a)) + a;
    return log1pf(expf(a - b)) + b;
}

/* denom[c] = -max - log(sum exp(x - max)) for every lattice cell c (row of V logits). */
static void denominators(const float* x, int64_t cells, int V, float* denom) {
    for (int64_t c = 0; c < cells; ++c) {
        const float* r = x + c * V;
        float m = r[0];
        for (int v = 1; v < V; ++v) m = r[v] > m ? r[v] : m;
        float s = 0.f;
        for (int v = 0; v < V; ++v) s += expf(r[v] - m);
        denom[c] = -m - logf(s);
    }
}

/*
 * logits  [B,T,U1,V] float32, contiguous     labels [B,U1-1] int64
 * flen    [B] valid T per utterance          glen   [B] valid label count (U_b = glen+1)
 * outputs: costs[B]; grads[B,T,U1,V] w.r.t. LOGITS (zero outside the valid lattice);
 *          alphas/betas [B,T,U1] (zero outside), grads_lp (optional) w.r.t. log-probs.
 * returns 0 on success, -1 on invalid sizes.
 */
int oracle_rnnt_loss(const float* logits, const int64_t* labels, const int64_t* flen, const int64_t* glen,
                     int B, int T, int U1, int V, int blank, float fastemit, float clamp,
                     float* costs, float* grads, float* alphas, float* betas, float* grads_lp) {
    if (B <= 0 || T <= 0 || U1 <= 0 || V <= 0 || blank < 0 || blank >= V) return -1;
    const int64_t cells = (int64_t)B * T * U1;
    float* denom = (float*)malloc(sizeof(float) * cells);
    if (!denom) return -1;
    denominators(logits, cells, V, denom);
    memset(alphas, 0, sizeof(float) * cells);
    memset(betas, 0, sizeof(float) * cells);
    if (grads) memset(grads, 0, sizeof(float) * cells * V);
    if (grads_lp) memset(grads_lp, 0, sizeof(float) * cells * V);
    const float l1p = log1pf(fastemit);
    for (int b = 0; b < B; ++b) {
        const int Tb = (int)flen[b], Ub = (int)glen[b] + 1;
        if (Tb < 1 || Tb > T || Ub < 1 || Ub > U1) { free(denom); return -1; }
        const int64_t base = (int64_t)b * T * U1;
        const int64_t* lab = labels + (int64_t)b * (U1 - 1);
#define C(t, u) (base + (int64_t)(t) * U1 + (u))
#define LP(t, u, v) (denom[C(t, u)] + logits[C(t, u) * V + (v)])
        float* a = alphas; float* be = betas;
        a[C(0, 0)] = 0.f;
        for (int t = 0; t < Tb; ++t)
            for (int u = 0; u < Ub; ++u) {
                if (u == 0 && t > 0) a[C(t, 0)] = a[C(t - 1, 0)] + LP(t - 1, 0, blank);
                if (t == 0 && u > 0) a[C(0, u)] = a[C(0, u - 1)] + LP(0, u - 1, lab[u - 1]);
                if (t > 0 && u > 0) {
                    float no_emit = a[C(t - 1, u)] + LP(t - 1, u, blank);
                    float emit = a[C(t, u - 1)] + LP(t, u - 1, lab[u - 1]);
                    a[C(t, u)] = lse2(emit, no_emit);
                }
            }
        const float ll = a[C(Tb - 1, Ub - 1)] + LP(Tb - 1, Ub - 1, blank);
        be[C(Tb - 1, Ub - 1)] = LP(Tb - 1, Ub - 1, blank);
        for (int t = Tb - 1; t >= 0; --t)
            for (int u = Ub - 1; u >= 0; --u) {
                if (u == Ub - 1 && t < Tb - 1) be[C(t, u)] = be[C(t + 1, u)] + LP(t, u, blank);
                if (t == Tb - 1 && u < Ub - 1) be[C(t, u)] = be[C(t, u + 1)] + LP(t, u, lab[u]);
                if (t < Tb - 1 && u < Ub - 1) {
                    float no_emit = be[C(t + 1, u)] + LP(t, u, blank);
                    float emit = be[C(t, u + 1)] + LP(t, u, lab[u]);
                    be[C(t, u)] = lse2(emit, no_emit);
                }
            }
        costs[b] = -(ll + ll * fastemit);
        for (int t = 0; t < Tb; ++t)
            for (int u = 0; u < Ub; ++u) {
                const int64_t c = C(t, u);
                if (grads_lp) {
                    if (t < Tb - 1)
                        grads_lp[c * V + blank] = -expf(LP(t, u, blank) + a[c] + be[C(t + 1, u)] - ll);
                    if (u < Ub - 1)
                        grads_lp[c * V + lab[u]] = -expf(l1p + LP(t, u, lab[u]) + a[c] + be[C(t, u + 1)] - ll);
                    if (t == Tb - 1 && u == Ub - 1) grads_lp[c * V + blank] = -expf(LP(t, u, blank) + a[c] - ll);
                }
                if (!grads) continue;
                for (int v = 0; v < V; ++v) {
                    const float logpk = LP(t, u, v);
                    float g = expf(a[c] + be[c] + logpk - ll);
                    if (fastemit > 0.f && u < Ub - 1)
                        g += fastemit * expf(a[c] + LP(t, u, lab[u]) + be[C(t, u + 1)] + logpk - ll);
                    if (v == blank && t == Tb - 1 && u == Ub - 1) g -= expf(a[c] + logpk - ll);
                    if (v == blank && t < Tb - 1) g -= expf(a[c] + logpk - ll + be[C(t + 1, u)]);
                    if (u < Ub - 1 && v == lab[u]) g -= expf(l1p + a[c] + logpk - ll + be[C(t, u + 1)]);
                    if (clamp > 0.f) { g = g < clamp ? g : clamp; g = g > -clamp ? g : -clamp; }
                    grads[c * V + v] = g;
                }
            }
#undef C
#undef LP
    }
    free(denom);
    return 0;
}

/*
 * CTC negative log-likelihood + gradient w.r.t. LOG-PROBS, restating torch.nn.CTCLoss(reduction='none',
 * zero_infinity=True) as the reference calls it (A/losses/ctc.py:45-82): log_probs [T,B,V] (time major),
 * targets [B,S] padded, blank index `blank`.  Gradient convention = ATen's: d loss_b / d log_probs[t,b,v]
 * = exp(lp) - exp(log sum_{s: l'_s = v} alpha_t(s) beta_t(s) + nll - lp), zero for t >= input_len.
 */
int oracle_ctc_loss(const float* lp, const int64_t* targets, const int64_t* in_len, const int64_t* tg_len,
                    int T, int B, int V, int S, int blank, int zero_infinity, float* nll, float* grad) {
    if (grad) memset(grad, 0, sizeof(float) * (size_t)T * B * V);
    for (int b = 0; b < B; ++b) {
        const int Tb = (int)in_len[b], Sb = (int)tg_len[b], L = 2 * Sb + 1;
        if (Tb < 0 || Tb > T || Sb < 0 || Sb > S) return -1;
        const int64_t* tg = targets + (int64_t)b * S;
        double* al = (double*)malloc(sizeof(double) * (size_t)(Tb > 0 ? Tb : 1) * L);
        double* be = (double*)malloc(sizeof(double) * (size_t)(Tb > 0 ? Tb : 1) * L);
        if (!al || !be) return -1;
#define LPI(t, v) ((double)lp[((int64_t)(t) * B + b) * V + (v)])
#define EXT(s) (((s) & 1) ? (int)tg[(s) >> 1] : blank)
        if (Tb == 0) { nll[b] = Sb == 0 ? 0.f : (zero_infinity ? 0.f : INFINITY); free(al); free(be); continue; }
        for (int s = 0; s < L; ++s) al[s] = -INFINITY;
        al[0] = LPI(0, blank);
        if (L > 1) al[1] = LPI(0, EXT(1));
        for (int t = 1; t < Tb; ++t)
            for (int s = 0; s < L; ++s) {
                double a0 = al[(t - 1) * L + s];
                double a1 = s >= 1 ? al[(t - 1) * L + s - 1] : -INFINITY;
                double a2 = (s >= 2 && EXT(s) != blank && EXT(s) != EXT(s - 2)) ? al[(t - 1) * L + s - 2] : -INFINITY;
                double m = a0 > a1 ? a0 : a1; m = m > a2 ? m : a2;
                al[t * L + s] = (m == -INFINITY) ? -INFINITY
                                : m + log(exp(a0 - m) + exp(a1 - m) + exp(a2 - m)) + LPI(t, EXT(s));
            }
        double l1 = al[(Tb - 1) * L + L - 1], l2 = L > 1 ? al[(Tb - 1) * L + L - 2] : -INFINITY;
        double m = l1 > l2 ? l1 : l2;
        double ll = (m == -INFINITY) ? -INFINITY : m + log(exp(l1 - m) + exp(l2 - m));
        double loss = -ll;
        int inf = isinf(loss);
        nll[b] = (inf && zero_infinity) ? 0.f : (float)loss;
        if (grad && !inf) {
            for (int s = 0; s < L; ++s) be[(Tb - 1) * L + s] = -INFINITY;
            be[(Tb - 1) * L + L - 1] = LPI(Tb - 1, blank);
            if (L > 1) be[(Tb - 1) * L + L - 2] = LPI(Tb - 1, EXT(L - 2));
            for (int t = Tb - 2; t >= 0; --t)
                for (int s = 0; s < L; ++s) {
                    double b0 = be[(t + 1) * L + s];
                    double b1 = s + 1 < L ? be[(t + 1) * L + s + 1] : -INFINITY;
                    double b2 = (s + 2 < L && EXT(s + 2) != blank && EXT(s) != EXT(s + 2)) ? be[(t + 1) * L + s + 2]
                                                                                           : -INFINITY;
                    double mm = b0 > b1 ? b0 : b1; mm = mm > b2 ? mm : b2;
                    be[t * L + s] = (mm == -INFINITY) ? -INFINITY
                                    : mm + log(exp(b0 - mm) + exp(b1 - mm) + exp(b2 - mm)) + LPI(t, EXT(s));
                }
            double* acc = (double*)malloc(sizeof(double) * V);
            for (int t = 0; t < Tb; ++t) {
                for (int v = 0; v < V; ++v) acc[v] = 0.0;
                for (int s = 0; s < L; ++s) {
                    double ab = al[t * L + s] + be[t * L + s];
                    if (ab != -INFINITY) acc[EXT(s)] += exp(ab + loss - LPI(t, EXT(s)));
                }
                for (int v = 0; v < V; ++v)
                    grad[((int64_t)t * B + b) * V + v] = (float)(exp(LPI(t, v)) - acc[v]);
            }
            free(acc);
        }
        free(al); free(be);
#undef LPI
#undef EXT
    }
    return 0;
}
