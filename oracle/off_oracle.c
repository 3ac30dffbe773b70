/*
 * ORACLE (C restatement) -- test infrastructure only; never linked into the product.
 *
 * Plain-C, double-accumulating restatement of the reference's OFF sub-network forward,
 * written from the algorithm (not from the PyTorch oracle's code): direct convolution
 * loops in NCHW, fp32 storage between layers exactly where the reference materialises
 * fp32 tensors, fp64 accumulation inside each dot product.  It is the independent
 * "truth" the fp32 GPU path and the PyTorch oracle are both compared against.
 *
 * Parity status: PINNED -- tests/test_oracle_c.py checks it against the goldens that
 * oracle/gen_golden.py captured from the reference import (tests/golden/*.npz).
 *
 * Reference lines (paths relative to the reference repository root):
 *   OFF unit            RGB_OFF.py:596-616 (gen 1x1 + ReLU :597-598, temporal diff :599-604,
 *                       flat slice :609, down 1x1 :610, depthwise 3x3 :611, concat :616);
 *                       Flow_OFF.py:622 + util.py:52-77 for the diagonal-Sobel variant
 *   fusion @28          RGB_OFF.py:655-685      fusion @14  :759-780      fusion @7  :831-841
 *   heads               RGB_OFF.py:782-787, 789-793, 843-847 (MaxPool(3,2,ceil) :353, AvgPool(7) :262)
 *   SegmentConsensus    basic_ops.py:19-21 as applied at Flow_OFF.py:867-876
 *
 * Weight order: the `weights` array follows offk_amd.spec.weight_shapes(variant):
 *   per site {gen.w, gen.b, down.w, down.b, [grad.w, grad.b]}; [sobel.w]; the 24 fusion
 *   convs {w, b} in forward-declaration order; the 3 heads {w, b} (7, 28, 14).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NSITES 9
static const int SITE_C[NSITES] = {256, 320, 576, 576, 576, 608, 608, 1024, 1024};
static const int SITE_H[NSITES] = {28, 28, 14, 14, 14, 14, 14, 7, 7};

static float* falloc(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }

/* y[n][co][ho][wo] = b[co] + sum_{ci,kh,kw} w[co][ci][kh][kw] * x[n][g*cig+ci][ho*s+kh-p][wo*s+kw-p]
 * (cross-correlation, zero padding -- nn.Conv2d).  groups == 1 or depthwise (groups == C). */
static void conv2d(const float* x, int n, int ci, int h, int w_, const float* wt, const float* b, int co, int k,
                   int s, int p, int depthwise, float* y) {
  const int ho = (h + 2 * p - k) / s + 1, wo = (w_ + 2 * p - k) / s + 1;
  const int cig = depthwise ? 1 : ci;
#pragma omp parallel for collapse(2) schedule(static)
  for (int in = 0; in < n; ++in)
    for (int oc = 0; oc < co; ++oc)
      for (int oy = 0; oy < ho; ++oy)
        for (int ox = 0; ox < wo; ++ox) {
          double acc = b ? (double)b[oc] : 0.0;
          for (int ic = 0; ic < cig; ++ic) {
            const int xc = depthwise ? oc : ic;
            const float* xp = x + ((size_t)in * ci + xc) * h * w_;
            const float* wp = wt + ((size_t)oc * cig + ic) * k * k;
            for (int ky = 0; ky < k; ++ky) {
              const int iy = oy * s + ky - p;
              if (iy < 0 || iy >= h) continue;
              for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s + kx - p;
                if (ix < 0 || ix >= w_) continue;
                acc += (double)wp[ky * k + kx] * (double)xp[(size_t)iy * w_ + ix];
              }
            }
          }
          y[(((size_t)in * co + oc) * ho + oy) * wo + ox] = (float)acc;
        }
}

static void relu_(float* x, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}
static void add_(float* a, const float* b, size_t n) {
  for (size_t i = 0; i < n; ++i) a[i] = a[i] + b[i];
}

/* MaxPool2d(3, stride 2, ceil_mode) then AvgPool2d(7) then Linear; x [n][c][h][h] */
static void head(const float* x, int n, int c, int h, int maxpool, const float* fw, const float* fb, int ncls,
                 float* out) {
  float* pooled = falloc((size_t)n * c);
  for (int in = 0; in < n; ++in)
    for (int ic = 0; ic < c; ++ic) {
      const float* xp = x + ((size_t)in * c + ic) * h * h;
      double s = 0.0;
      if (maxpool) {
        const int ho = (h - 3 + 1) / 2 + 1;
        for (int oy = 0; oy < ho; ++oy)
          for (int ox = 0; ox < ho; ++ox) {
            float m = -INFINITY;
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                int iy = 2 * oy + ky, ix = 2 * ox + kx;
                if (iy < h && ix < h && xp[iy * h + ix] > m) m = xp[iy * h + ix];
              }
            s += (double)m;
          }
        s /= (double)(ho * ho);
      } else {
        for (int i = 0; i < h * h; ++i) s += (double)xp[i];
        s /= (double)(h * h);
      }
      pooled[(size_t)in * c + ic] = (float)s;
    }
  for (int in = 0; in < n; ++in)
    for (int o = 0; o < ncls; ++o) {
      double a = (double)fb[o];
      for (int ic = 0; ic < c; ++ic) a += (double)fw[(size_t)o * c + ic] * (double)pooled[(size_t)in * c + ic];
      out[(size_t)in * ncls + o] = (float)a;
    }
  free(pooled);
}

static void consensus(const float* x, int B, int T, int C, float* out) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int t = 0; t < T; ++t) s += (double)x[((size_t)b * T + t) * C + c];
      out[(size_t)b * C + c] = (float)(s / (double)T);
    }
}

/* copy src [n][c][hw] into channels [coff, coff+c) of dst [n][ctot][hw] (torch.cat, dim 1) */
static void put_channels(float* dst, int ctot, int coff, const float* src, int n, int c, int hw) {
  for (int in = 0; in < n; ++in)
    memcpy(dst + ((size_t)in * ctot + coff) * hw, src + (size_t)in * c * hw, (size_t)c * hw * sizeof(float));
}

typedef struct { const float* const* w; int i; } wcur;
static const float* nextw(wcur* c) { return c->w[c->i++]; }

/* one bottleneck step: y = conv(x); optional relu */
static float* conv_new(const float* x, int n, int ci, int h, wcur* wc, int co, int k, int s, int p, int relu,
                       int* ho_out) {
  const float* w = nextw(wc);
  const float* b = nextw(wc);
  const int ho = (h + 2 * p - k) / s + 1;
  float* y = falloc((size_t)n * co * ho * ho);
  conv2d(x, n, ci, h, h, w, b, co, k, s, p, 0, y);
  if (relu) relu_(y, (size_t)n * co * ho * ho);
  if (ho_out) *ho_out = ho;
  return y;
}

int off_oracle_forward(const float* const feats[NSITES], const float* const* weights, int B, int L, int variant,
                       int slice_mode, int do_consensus, int ncls, float* out7, float* out14, float* out28,
                       float* fusion28_out, float* fusion14_out, float* fusion7_out, float* sum7_out) {
  const int N = B * L, T = L - 1, P = B * T;
  wcur wc = {weights, 0};
  float* F28 = falloc((size_t)P * 320 * 784);
  float* F14 = falloc((size_t)P * 1056 * 196);
  float* F7 = falloc((size_t)P * 832 * 49);
  const float* sobel = NULL;
  /* the shared sobel weight sits after the nine sites in the weight order */
  if (variant == 1) sobel = weights[NSITES * 4];
  static const int fus_of[NSITES] = {0, 0, 1, 1, 1, 1, 1, 2, 2};
  static const int coff_of[NSITES] = {0, 160, 0, 160, 320, 480, 640, 0, 160};
  for (int s = 0; s < NSITES; ++s) {
    const int C = SITE_C[s], H = SITE_H[s], HW = H * H;
    const float *gw = nextw(&wc), *gb = nextw(&wc), *dw = nextw(&wc), *db = nextw(&wc);
    const float *sw = NULL, *sb = NULL;
    if (variant == 0) { sw = nextw(&wc); sb = nextw(&wc); } else { sw = sobel; }
    float* G = falloc((size_t)N * 128 * HW);
    conv2d(feats[s], N, C, H, H, gw, gb, 128, 1, 1, 0, 0, G);
    relu_(G, (size_t)N * 128 * HW);
    /* temporal: pair (b,t) = frame (b,t+1) - frame (b,t) */
    float* M = falloc((size_t)P * 160 * HW);
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < T; ++t) {
        const float* f1 = G + (size_t)(b * L + t + 1) * 128 * HW;
        const float* f0 = G + (size_t)(b * L + t) * 128 * HW;
        float* m = M + ((size_t)(b * T + t) * 160 + 32) * HW;
        for (size_t i = 0; i < (size_t)128 * HW; ++i) m[i] = f1[i] - f0[i];
      }
    /* spatial: source frame of output pair p */
    float* Xs = falloc((size_t)P * C * HW);
    for (int p = 0; p < P; ++p) {
      int f = slice_mode == 0 ? p : (p / T) * L + (p % T);
      memcpy(Xs + (size_t)p * C * HW, feats[s] + (size_t)f * C * HW, (size_t)C * HW * sizeof(float));
    }
    float* D = falloc((size_t)P * 32 * HW);
    conv2d(Xs, P, C, H, H, dw, db, 32, 1, 1, 0, 0, D);
    float* S = falloc((size_t)P * 32 * HW);
    conv2d(D, P, 32, H, H, sw, sb, 32, 3, 1, 1, 1, S);
    put_channels(M, 160, 0, S, P, 32, HW);
    float* F = fus_of[s] == 0 ? F28 : fus_of[s] == 1 ? F14 : F7;
    const int ctot = fus_of[s] == 0 ? 320 : fus_of[s] == 1 ? 1056 : 832;
    put_channels(F, ctot, coff_of[s], M, P, 160, HW);
    free(G); free(M); free(Xs); free(D); free(S);
  }
  if (variant == 1) nextw(&wc); /* skip sobel */

  /* ---- fusion @28 ---- */
  int h14;
  float* x0 = conv_new(F28, P, 320, 28, &wc, 64, 7, 2, 3, 0, &h14);          /* pre-ReLU kept for the branch */
  const size_t n64 = (size_t)P * 64 * 196, n256 = (size_t)P * 256 * 196;
  float* a = falloc(n64);
  memcpy(a, x0, n64 * sizeof(float));
  relu_(a, n64);
  float* t1 = conv_new(a, P, 64, 14, &wc, 64, 1, 1, 0, 1, NULL);
  float* t2 = conv_new(t1, P, 64, 14, &wc, 64, 3, 1, 1, 1, NULL);
  float* c3 = conv_new(t2, P, 64, 14, &wc, 256, 1, 1, 0, 0, NULL);
  float* br = conv_new(x0, P, 64, 14, &wc, 256, 1, 1, 0, 0, NULL);
  add_(c3, br, n256);
  relu_(c3, n256);
  float* s28 = c3;
  free(a); free(t1); free(t2); free(br); free(x0);
  for (int blk = 0; blk < 2; ++blk) {
    t1 = conv_new(s28, P, 256, 14, &wc, 64, 1, 1, 0, 1, NULL);
    t2 = conv_new(t1, P, 64, 14, &wc, 64, 3, 1, 1, 1, NULL);
    c3 = conv_new(t2, P, 64, 14, &wc, 256, 1, 1, 0, 0, NULL);
    add_(c3, s28, n256);
    relu_(c3, n256);
    free(t1); free(t2); free(s28);
    s28 = c3;
  }
  put_channels(F14, 1056, 800, s28, P, 256, 196);

  /* ---- fusion @14 ---- */
  const size_t n512 = (size_t)P * 512 * 49;
  float* x1 = conv_new(F14, P, 1056, 14, &wc, 128, 5, 2, 2, 1, NULL);
  t1 = conv_new(x1, P, 128, 7, &wc, 128, 1, 1, 0, 1, NULL);
  t2 = conv_new(t1, P, 128, 7, &wc, 128, 3, 1, 1, 1, NULL);
  c3 = conv_new(t2, P, 128, 7, &wc, 512, 1, 1, 0, 0, NULL);
  float* ex = conv_new(x1, P, 128, 7, &wc, 512, 1, 1, 0, 0, NULL);
  add_(c3, ex, n512);
  relu_(c3, n512);
  float* s14 = c3;
  free(t1); free(t2); free(ex); free(x1);
  t1 = conv_new(s14, P, 512, 7, &wc, 128, 1, 1, 0, 1, NULL);
  t2 = conv_new(t1, P, 128, 7, &wc, 128, 3, 1, 1, 1, NULL);
  c3 = conv_new(t2, P, 128, 7, &wc, 512, 3, 1, 1, 1, NULL);                 /* 3x3 with its own ReLU */
  add_(c3, s14, n512);
  relu_(c3, n512);
  free(t1); free(t2); free(s14);
  float* s14b = c3;
  put_channels(F7, 832, 320, s14b, P, 512, 49);

  /* ---- fusion @7 ---- */
  const size_t n1024 = (size_t)P * 1024 * 49;
  float* x2 = conv_new(F7, P, 832, 7, &wc, 256, 3, 1, 1, 1, NULL);
  t1 = conv_new(x2, P, 256, 7, &wc, 256, 1, 1, 0, 1, NULL);
  t2 = conv_new(t1, P, 256, 7, &wc, 256, 3, 1, 1, 1, NULL);
  c3 = conv_new(t2, P, 256, 7, &wc, 1024, 1, 1, 0, 0, NULL);
  br = conv_new(x2, P, 256, 7, &wc, 1024, 1, 1, 0, 0, NULL);
  add_(c3, br, n1024);                                                       /* no ReLU (RGB_OFF.py:841) */
  float* s7 = c3;
  free(t1); free(t2); free(br); free(x2);

  /* ---- heads: weight order is (7, 28, 14) ---- */
  const float *w7 = nextw(&wc), *b7 = nextw(&wc), *w28 = nextw(&wc), *b28 = nextw(&wc), *w14 = nextw(&wc), *b14 = nextw(&wc);
  float* l7 = falloc((size_t)P * ncls);
  float* l14 = falloc((size_t)P * ncls);
  float* l28 = falloc((size_t)P * ncls);
  head(s7, P, 1024, 7, 0, w7, b7, ncls, l7);
  head(s14b, P, 512, 7, 0, w14, b14, ncls, l14);
  head(s28, P, 256, 14, 1, w28, b28, ncls, l28);
  if (do_consensus) {
    consensus(l7, B, T, ncls, out7);
    consensus(l14, B, T, ncls, out14);
    consensus(l28, B, T, ncls, out28);
  } else {
    memcpy(out7, l7, (size_t)P * ncls * sizeof(float));
    memcpy(out14, l14, (size_t)P * ncls * sizeof(float));
    memcpy(out28, l28, (size_t)P * ncls * sizeof(float));
  }
  if (fusion28_out) memcpy(fusion28_out, F28, (size_t)P * 320 * 784 * sizeof(float));
  if (fusion14_out) memcpy(fusion14_out, F14, (size_t)P * 1056 * 196 * sizeof(float));
  if (fusion7_out) memcpy(fusion7_out, F7, (size_t)P * 832 * 49 * sizeof(float));
  if (sum7_out) memcpy(sum7_out, s7, n1024 * sizeof(float));
  free(l7); free(l14); free(l28); free(s7); free(s14b); free(s28); free(F28); free(F14); free(F7);
  return 0;
}
