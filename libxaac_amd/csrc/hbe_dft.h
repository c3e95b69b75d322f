/*
 * hbe_dft.h -- the DFT-domain harmonic transposer of the eSBR tool (-esbr_hq:1) as host/device code:
 * ixheaacd_dft_hbe_apply (decoder/ixheaacd_hbe_dft_trans.c:771-941) with its polar helpers (ixheaacd_karth2polar :457,
 * ixheaacd_dft_hbe_apply_polar_t2 / _t3 / _t :576 / :635 / :712).  Included by the oracle (oracle/oracle_hbe.cpp, one
 * "lane") and by the kernel (hbe_kernel.hip: a 256-thread workgroup per channel-frame), both compiled without
 * floating-point contraction.
 *
 * A frame is eight hops.  Each hop windows ana_fft_size[0] samples of the sub-sampled core-band signal, transforms them
 * (real FFT), takes magnitude and phase, writes the stretched spectra of the factors 2 .. max_stretch into a second
 * spectrum through the patches' cross-over windows, transforms that back and overlap-adds it into the output signal the
 * analysis bank (hbe_poly.h: xh_dft_anal_band) then splits into QMF sub-bands.  Every expression of the polar part keeps
 * the reference's operand order, widths and libm calls (double-precision atan2 / sqrt / cos / sin / pow / cbrt on float
 * arguments, rounded to float where the reference rounds).  The transforms do NOT follow the reference's
 * (ixheaacd_fft_ifft_32x32.c:252-1587: hand-unrolled radix-4 / radix-3 / radix-7 code per size): a half-length complex
 * transform of N = 16 N2 points runs as two passes of short sums (16-point sums, a twiddle, N2-point sums) that spread
 * over the lanes, with twiddles computed in double precision; the (sin, cos) pairs of the real-transform step are the
 * reference's ROM (six-decimal values, tables_hbe.inc).  The two differ by float rounding, and so does everything behind
 * them: see include/xaac_hbe.h for the tolerance the tests hold.
 */
#ifndef XAAC_HBE_DFT_H
#define XAAC_HBE_DFT_H

#include <math.h>

#include "hbe_poly.h"

#pragma clang fp contract(off)

#ifndef XD_T
#define XD_T(i) /* stage timers of profiling builds (hbe_kernel.hip, tools/prof_hbe_dft.py) */
#endif

struct XdC {
  float r, i;
};

struct XdSizes {
  int s, L;              /* synth_size, analy_size */
  int ana0, ana;         /* ana_fft_size[0], ana_fft_size[oversampling] */
  int syn0, syn;         /* syn_fft_size[0], syn_fft_size[oversampling] */
  int fft;               /* fft_size[oversampling]: 1024 / 1536 */
  int in_hop, out_hop, ana_pad, syn_pad, ana_off, syn_off;
  int ms;                /* max_stretch */
};

/* the sizes of a frame (hbe_dft_trans.c:289-324, :778-798); false where the reference has no transform for them
   (ixheaacd_hbe_fft_map, :508-549) or a buffer would not hold them */
FX_HD bool xd_sizes(const xaac_hbe_dft_state *st, int ovs, XdSizes *z) {
  const int s = st->synth_size, L = st->anal.analy_size, ks = st->k_start, a0 = st->anal.a_start;
  if (s != 8 && s != 12 && s != 16) return false;
  if (L < 4 || L > 48 || (L & 3)) return false;
  z->s = s;
  z->L = L;
  z->ana0 = 32 * s;
  z->ana = ovs ? 48 * s : 32 * s;
  z->syn0 = 16 * L;
  z->syn = ovs ? 24 * L : 16 * L;
  z->fft = ovs ? 1536 : 1024;
  if (z->ana != 384 && z->ana != 512 && z->ana != 576 && z->ana != 768) return false;
  if (z->syn != 448 && z->syn != 512 && z->syn != 672 && z->syn != 768) return false;
  z->in_hop = z->ana0 / 8;
  z->out_hop = 2 * z->syn0 / 8;
  z->ana_pad = (z->ana - z->ana0) / 2;
  z->syn_pad = (z->syn - z->syn0) / 2;
  z->ana_off = ks * z->fft / 32;
  z->syn_off = a0 * z->fft / 64;
  z->ms = st->max_stretch;
  if (ks < 0 || ks + s > 64 || ks * 32 + 2 * s > 7 * 64) return false; /* the synthesis bank's modulation table */
  if (a0 < 0 || a0 + L > 64) return false;
  if ((z->ana_off & 1) || (z->syn_off & 1)) return false; /* (pairs of floats are transformed as complex words) */
  if (z->ana_off + z->ana > z->fft || z->syn_off + z->syn > z->fft) return false;
  if (z->ms < 0 || z->ms > 4) return false;
  return true;
}

FX_HD const float *xd_rot(int len) { /* ixheaacd_hbe_fft_map's ana_cos_sin_tab / syn_cos_sin_tab: (sin, cos)(2 pi i / len), i = 1 .. len / 4 */
  switch (len) {
    case 384: return xaac_hbe_dft_rot_384;
    case 448: return xaac_hbe_dft_rot_448;
    case 512: return xaac_hbe_dft_rot_512;
    case 576: return xaac_hbe_dft_rot_576;
    case 672: return xaac_hbe_dft_rot_672;
    default: return xaac_hbe_dft_rot_768;
  }
}

/* w[j] = exp(-2 pi i j / n), j < n */
template <class CX>
FX_HD void xd_twiddles(const CX &cx, XdC *w, int n) {
  for (int j = cx.lane; j < n; j += cx.n) {
    const double a = 6.283185307179586476925286766559 * (double)j / (double)n;
    w[j].r = (float)cos(a);
    w[j].i = (float)-sin(a);
  }
}

/* In-place complex transform of n = 16 n2 points, X[k] = sum x[m] exp(sign 2 pi i m k / n), not scaled; w from xd_twiddles,
   tmp: n words.  m = n2 m1 + m2, k = k1 + 16 k2:  W_n^(m k) = W_16^(m1 k1) W_n^(m2 k1) W_n2^(m2 k2). */
template <class CX>
FX_HD void xd_cfft(const CX &cx, XdC *x, XdC *tmp, const XdC *w, int n, int sign) {
  const int n2 = n / 16;
  for (int e = cx.lane; e < n; e += cx.n) {
    const int m2 = e >> 4, k1 = e & 15;
    float ar = 0.0f, ai = 0.0f;
    for (int m1 = 0; m1 < 16; m1++) {
      const XdC v = x[n2 * m1 + m2], t = w[((m1 * k1) & 15) * n2];
      const float ti = sign < 0 ? t.i : -t.i;
      ar += v.r * t.r - v.i * ti;
      ai += v.r * ti + v.i * t.r;
    }
    const XdC t = w[m2 * k1];
    const float ti = sign < 0 ? t.i : -t.i;
    tmp[e].r = ar * t.r - ai * ti;
    tmp[e].i = ar * ti + ai * t.r;
  }
  cx.sync();
  for (int e = cx.lane; e < n; e += cx.n) {
    const int k2 = e >> 4, k1 = e & 15;
    float ar = 0.0f, ai = 0.0f;
    int q = 0; /* (m2 k2) mod n2 */
    for (int m2 = 0; m2 < n2; m2++) {
      const XdC v = tmp[16 * m2 + k1], t = w[16 * q];
      const float ti = sign < 0 ? t.i : -t.i;
      ar += v.r * t.r - v.i * ti;
      ai += v.r * ti + v.i * t.r;
      q += k2;
      if (q >= n2) q -= n2;
    }
    x[k1 + 16 * k2].r = ar;
    x[k1 + 16 * k2].i = ai;
  }
  cx.sync();
}

/* :830-856: the half-length transform of the packed real signal becomes the real signal's spectrum (bins 0 .. len / 2, the
   last one's real part in word 1) */
template <class CX>
FX_HD void xd_real_post(const CX &cx, float *d, int len) {
  const float *cs = xd_rot(len);
  for (int i = cx.lane; i <= len / 4; i += cx.n) {
    if (i == 0) {
      const float t = d[0] + d[1];
      d[1] = d[0] - d[1];
      d[0] = t;
      continue;
    }
    const float c = cs[2 * (i - 1)], sn = cs[2 * (i - 1) + 1];
    float tmp1 = d[2 * i] - d[len - 2 * i];
    float tmp2 = d[2 * i + 1] + d[len - 2 * i + 1];
    const float tmp3 = c * tmp1 - sn * tmp2;
    const float tmp4 = sn * tmp1 + c * tmp2;
    tmp1 = d[2 * i] + d[len - 2 * i];
    tmp2 = d[2 * i + 1] - d[len - 2 * i + 1];
    d[2 * i + 0] = 0.5f * (tmp1 - tmp3);
    d[2 * i + 1] = 0.5f * (tmp2 - tmp4);
    d[len - 2 * i + 0] = 0.5f * (tmp1 + tmp3);
    d[len - 2 * i + 1] = -0.5f * (tmp2 + tmp4);
  }
  cx.sync();
}

/* :898-926: the way back, scaled by 1 / len */
template <class CX>
FX_HD void xd_real_pre(const CX &cx, float *d, int len) {
  const float *cs = xd_rot(len);
  const float scale = 1.0f / len;
  for (int i = cx.lane; i <= len / 4; i += cx.n) {
    if (i == 0) {
      const float t = d[0] + d[1];
      d[1] = scale * (d[0] - d[1]);
      d[0] = scale * t;
      continue;
    }
    const float c = cs[2 * (i - 1)], sn = cs[2 * (i - 1) + 1];
    float tmp1 = d[2 * i] - d[len - 2 * i];
    float tmp2 = d[2 * i + 1] + d[len - 2 * i + 1];
    const float tmp3 = c * tmp1 + sn * tmp2;
    const float tmp4 = -sn * tmp1 + c * tmp2;
    tmp1 = d[2 * i] + d[len - 2 * i];
    tmp2 = d[2 * i + 1] - d[len - 2 * i + 1];
    d[2 * i] = scale * (tmp1 - tmp3);
    d[2 * i + 1] = scale * (tmp2 - tmp4);
    d[len - 2 * i] = scale * (tmp1 + tmp3);
    d[len - 2 * i + 1] = -scale * (tmp2 + tmp4);
  }
  cx.sync();
}

/* ixheaacd_karth2polar (:457-481): mag / phase of bins 0 .. fft_size / 2 of the packed spectrum */
template <class CX>
FX_HD void xd_polar(const CX &cx, const float *sp, float *mag, float *phase, int fft_size) {
  for (int m = cx.lane; m <= fft_size / 2; m += cx.n) {
    if (m == 0 || m == fft_size / 2) {
      const float v = sp[m == 0 ? 0 : 1];
      if (v < 0) {
        phase[m] = (float)acos(-1.0);
        mag[m] = -v;
      } else {
        phase[m] = 0;
        mag[m] = v;
      }
    } else {
      phase[m] = (float)atan2((double)sp[2 * m + 1], (double)sp[2 * m]);
      mag[m] = (float)sqrt((double)(sp[2 * m] * sp[2 * m] + sp[2 * m + 1] * sp[2 * m + 1]));
    }
  }
  cx.sync();
}

#define XD_MIN(a, b) ((a) < (b) ? (a) : (b))
FX_HD void xd_sincos(double x, double *s, double *c) { sincos(x, s, c); }

/* The pitch-adaptive search all three variants share (:612-623, :681-692, :744-755): the source bin pair whose weaker magnitude
   is largest.  Returns m_val; m_tr / utk are written when a pair was found. */
FX_HD float xd_cross_search(const float *mag, int i, int T, int p, float p_flt, int fft_size, int *m_tr, int *utk) {
  float m_val = 0;
  for (int tr = 1; tr < T; tr++) {
    const int ti = (int)((2.0f * i - tr * p_flt) / T + 0.5f);
    if ((ti < 0) || (ti + p > fft_size / 2)) continue;
    const float temp = XD_MIN(mag[ti], mag[ti + p]);
    if (temp > m_val) {
      m_val = temp;
      *m_tr = tr;
      *utk = ti;
    }
  }
  return m_val;
}

/* Bin i of the transposed spectrum: the contributions of the factors 2 .. ms in the reference's order (:871-890).  win(T):
   fd_win_buf[T - 2][oversampling][i].  A factor whose window is zero at this bin adds (signed) zeros: left out. */
template <class WIN>
FX_HD void xd_transpose_bin(const float *mag, const float *phase, const WIN &win, int i, int ms, int pitch_in_bins, int fft_size,
                            float *out_r, float *out_i) {
  float sr = 0.0f, si = 0.0f;
  const float p_flt = fft_size * pitch_in_bins / 1536.0f;
  const int p = (int)p_flt;
  const float q_thr = 4.0f;
  for (int T = 2; T <= ms; T++) {
    const float w = win(T);
    if (w == 0.0f) continue;
    float mag_t = 0, phase_t;
    double sn, cs; /* sin and cos of one argument come from one call (the values of the two separate calls the reference makes) */
    int m_tr = 0;
    if (T == 2) { /* ixheaacd_dft_hbe_apply_polar_t2 */
      int utk = i;
      mag_t = w * mag[utk];
      phase_t = T * phase[utk];
      if (phase_t == 0.0) {
        sr += mag_t;
      } else {
        xd_sincos((double)phase_t, &sn, &cs);
        sr += mag_t * (float)cs;
        si += mag_t * (float)sn;
      }
      if (p > 0) {
        const float m_val = xd_cross_search(mag, i, T, p, p_flt, fft_size, &m_tr, &utk);
        if (m_val > q_thr * mag[2 * i / T]) {
          mag_t = (float)((double)w * sqrt((double)mag[utk]) * sqrt((double)mag[utk + p]));
          phase_t = (T - m_tr) * phase[utk] + m_tr * phase[utk + p];
          xd_sincos((double)phase_t, &sn, &cs);
          sr += (float)((double)mag_t * cs);
          si += (float)((double)mag_t * sn);
        }
      }
    } else if (T == 3) { /* ixheaacd_dft_hbe_apply_polar_t3 */
      int utk = 2 * i / T;
      const float ptk = (2.0f * i / T) - utk;
      float k;
      if (i % 3 == 0) {
        mag_t = w * mag[utk];
      } else if (i % 3 == 1) {
        k = (float)cbrt((double)mag[utk]);
        mag_t = w * k * (float)pow((double)mag[utk + 1], (double)ptk);
      } else {
        k = (float)cbrt((double)mag[utk + 1]);
        mag_t = w * (float)pow((double)mag[utk], 1.0 - (double)ptk) * k;
      }
      phase_t = T * ((1 - ptk) * phase[utk] + ptk * phase[utk + 1]);
      xd_sincos((double)phase_t, &sn, &cs);
      sr += mag_t * (float)cs;
      si += mag_t * (float)sn;
      if (p > 0) {
        const float m_val = xd_cross_search(mag, i, T, p, p_flt, fft_size, &m_tr, &utk);
        if (m_val > q_thr * mag[2 * i / T]) {
          const float r = (float)m_tr / T;
          if (m_tr == 1) {
            k = (float)(cbrt((double)(float)mag[utk + p]));
            mag_t = w * (float)pow((double)mag[utk], 1.0 - (double)r) * k;
            phase_t = (T - m_tr) * phase[utk] + phase[utk + p];
          } else if (m_tr == 2) {
            k = (float)(cbrt((double)(float)mag[utk]));
            mag_t = w * k * (float)pow((double)mag[utk + p], (double)r);
            phase_t = phase[utk] + m_tr * phase[utk + p];
          }
          xd_sincos((double)phase_t, &sn, &cs);
          sr += mag_t * (float)cs;
          si += mag_t * (float)sn;
        }
      }
    } else { /* ixheaacd_dft_hbe_apply_polar_t */
      int utk = 2 * i / T;
      const float ptk = (2.0f * i / T) - utk;
      if (ptk == 0.0f) /* every second bin: pow(x, 1) is x and pow(y, 0) is 1, exactly (C99 F.9.4.4) -- the two calls left out */
        mag_t = w * mag[utk] * 1.0f;
      else
        mag_t = w * (float)pow((double)mag[utk], (double)(1.0f - ptk)) * (float)pow((double)mag[utk + 1], (double)ptk);
      phase_t = T * ((1 - ptk) * phase[utk] + ptk * phase[utk + 1]);
      xd_sincos((double)phase_t, &sn, &cs);
      sr += mag_t * (float)cs;
      si += mag_t * (float)sn;
      if (p > 0) {
        const float m_val = xd_cross_search(mag, i, T, p, p_flt, fft_size, &m_tr, &utk);
        if (m_val > q_thr * mag[2 * i / T]) {
          const float r = (float)m_tr / T;
          mag_t = w * (float)pow((double)mag[utk], 1.0 - (double)r) * (float)pow((double)mag[utk + p], (double)r);
          phase_t = (T - m_tr) * phase[utk] + m_tr * phase[utk + p];
          xd_sincos((double)phase_t, &sn, &cs);
          sr += mag_t * (float)cs;
          si += mag_t * (float)sn;
        }
      }
    }
  }
  *out_r = sr;
  *out_i = si;
}

/* the arrays a channel-frame works in (LDS on the GPU) */
struct XdWork {
  float *in;    /* [2 ana0]  ptr_input_buf */
  float *out;   /* [4 syn0]  ptr_output_buf */
  float *spec;  /* [ana]     ptr_spectrum + ana_fft_offset: the words the transform works on */
  float *awin;  /* [ana0]    anal_window */
  float *tx;    /* [fft + 2] ptr_spectrum_tx */
  float *mag;   /* [fft / 2 + 2] */
  float *phase; /* [fft / 2 + 2] */
  XdC *wa, *ws; /* [ana / 2], [syn / 2] twiddles */
  XdC *tmp;     /* [max(ana, syn) / 2] */
};

/* The eight hops (:814-935) on the signals in w->in / w->out (input already shifted and filled by the synthesis bank, output
   already shifted and its upper half cleared).  Of ptr_spectrum only the ana_fft_size words behind ana_fft_offset are ever read
   (the transform, then ixheaacd_karth2polar), so only those are made -- with the halves already in the places
   ixheaacd_dft_hbe_fft_memmove gives them; mag / phase outside that range stay the zeros they start as; the way back reads the
   transform's output through the same exchange of halves.  w->awin: the analysis window where the hops read it fastest. */
template <class CX>
FX_HD void xd_hops(const CX &cx, const XdSizes &z, const xaac_hbe_dft_cfg *cfg, int ovs, int pitch_in_bins, const XdWork *w) {
  xd_twiddles(cx, w->wa, z.ana / 2);
  xd_twiddles(cx, w->ws, z.syn / 2);
  const int half = z.fft / 2;
  for (int e = cx.lane; e < half + 2; e += cx.n) {
    w->mag[e] = 0.0f;
    w->phase[e] = 0.0f;
  }
  for (int e = cx.lane; e < z.ana0; e += cx.n) w->awin[e] = cfg->anal_window[e];
  cx.sync();
  XD_T(0);
  for (int hop = 0; hop < 8; hop++) {
    const float *src = w->in + hop * z.in_hop;
    float *a = w->spec; /* = ptr_spectrum + ana_fft_offset */
    /* :815-826: word r of the range comes from word r +- ana / 2 of the windowed, padded block */
    for (int r = cx.lane; r < z.ana; r += cx.n) {
      const int j = (r < z.ana / 2 ? r + z.ana / 2 : r - z.ana / 2) - z.ana_pad;
      a[r] = (j >= 0 && j < z.ana0) ? src[j] * w->awin[j] : 0.0f;
    }
    cx.sync();
    XD_T(1);
    xd_cfft(cx, reinterpret_cast<XdC *>(a), w->tmp, w->wa, z.ana / 2, -1);
    xd_real_post(cx, a, z.ana);
    XD_T(2);
    xd_polar(cx, a, w->mag + z.ana_off / 2, w->phase + z.ana_off / 2, z.ana);
    XD_T(3);
    /* :862-891 (ptr_spectrum_tx has fft_size + 2 words) */
    for (int i = cx.lane; i <= half; i += cx.n) {
      float r, im;
      xd_transpose_bin(w->mag, w->phase, [&](int T) { return cfg->fd_win[T - 2][ovs][i]; }, i, z.ms, pitch_in_bins, z.fft, &r, &im);
      w->tx[2 * i] = r;
      w->tx[2 * i + 1] = im;
    }
    cx.sync();
    XD_T(4);
    float *b = w->tx + z.syn_off;
    if (cx.lane == 0) b[1] = b[z.syn]; /* :893 */
    cx.sync();
    xd_real_pre(cx, b, z.syn);
    xd_cfft(cx, reinterpret_cast<XdC *>(b), w->tmp, w->ws, z.syn / 2, 1);
    float *dst = w->out + hop * z.out_hop;
    for (int j = cx.lane; j < z.syn0; j += cx.n) { /* :928-937 */
      const int r = z.syn_pad + j;
      dst[j] += b[r < z.syn / 2 ? r + z.syn / 2 : r - z.syn / 2] * cfg->synth_window[j];
    }
    cx.sync();
    XD_T(5);
  }
}

#endif /* XAAC_HBE_DFT_H */
