/* CPU oracle, second statement, plain C (C99, double precision).  TEST INFRASTRUCTURE ONLY.
 *
 * An independent restatement of the hot path of mauro-belgiovine/DL-channel-estimation-MaMIMO for the tests: it shares no code with
 * oracle/csi_oracle.py (no FFT library, no BLAS: a direct 256-point DFT and plain dot products), so an error in one statement does
 * not hide in the other.  Only tests/, __graft_entry__ (build(): compiling the checker is not using it) and bench.py's cpu_baseline
 * leg may touch anything under oracle/; the product library never links or loads this file.
 *
 * Pin status, as in csi_oracle.py: oc_ofdm_demod is PINNED against the spectra the reference's own numpy code computed
 * (tests/golden/ref_ofdm_reshape_nt4.npz, massiveMIMO_dataGenerator.py:425-453); the despread and the Dense / BatchNormalization
 * arithmetic are restated from helperMIMOChannelEstimate.m:8-41 and massiveMIMO_CSI_prediction_DNN.py:176-234 (MATLAB / TensorFlow
 * are not in this image: PARITY UNPINNED for those two, checked by identities and against the numpy statement).
 *
 * Paths in the comments are relative to the reference repository.  Build: gcc -O3 -std=c99 -shared -fPIC csi_oracle_c.c -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { OC_FFT = 256, OC_CP = 64, OC_SYM = 320, OC_NDATA = 234 };   /* generate_maMIMO_LTF.m:96-98 */

/* 1-based bin k is a data carrier unless it is a guard / DC bin ([1:7 129 251:256], generate_maMIMO_LTF.m:99) or one of the
   eight pilot bins (generate_maMIMO_LTF.m:100); generate_maMIMO_LTF.m:101-102 */
static int oc_is_data_bin(int k)
{
    static const int pilots[8] = {26, 54, 90, 118, 140, 168, 204, 232};
    if (k <= 7 || k == 129 || k >= 251) return 0;
    for (int i = 0; i < 8; ++i) if (k == pilots[i]) return 0;
    return 1;
}

/* out[234]: the 1-based data bins in ascending order; returns their number */
int oc_data_bins(int* out)
{
    int n = 0;
    for (int k = 1; k <= OC_FFT; ++k) if (oc_is_data_bin(k)) out[n++] = k;
    return n;
}

/* The 256-entry VHT-LTF literal of helperMIMOChannelEstimate.m:16-23, entry 0 = MATLAB index 1 (most negative frequency) */
void oc_vht_ltf_256(double* seq)
{
    static const signed char left[26]  = {1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1};
    static const signed char right[26] = {1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1};
    static const signed char mid_a[11] = {-1, -1, -1, 1, 1, -1, 1, -1, 1, 1, -1};
    static const signed char mid_b[9]  = {1, -1, 1, -1, 0, 1, -1, -1, 1};
    int n = 0;
    for (int i = 0; i < 7; ++i) seq[n++] = 0.0;
    for (int blk = 0; blk < 4; ++blk) {
        for (int i = 0; i < 26; ++i) seq[n++] = left[i];
        seq[n++] = 1.0;
        for (int i = 0; i < 26; ++i) seq[n++] = right[i];
        if (blk == 0 || blk == 2) for (int i = 0; i < 11; ++i) seq[n++] = mid_a[i];
        if (blk == 1)             for (int i = 0; i < 9; ++i)  seq[n++] = mid_b[i];
    }
    for (int i = 0; i < 6; ++i) seq[n++] = 0.0;
    /* n == 256 by construction: 7 + 4*53 + 11 + 9 + 11 + 6 */
}

/* a-1.  ltf [n_items][320*nt] (one rx antenna's preamble per item, re / im planes) -> rxsym [n_items][234][nt] (bins x symbol).
   Symbol s = samples 320 s .. 320 s + 319 (massiveMIMO_dataGenerator.py:437-439), window = samples 64..319 of it (:442-443),
   un-scaled 256-point DFT (:452), DC on 1-based bin 129 (generate_maMIMO_LTF.m:99), the 234 data bins kept (:98-102). */
void oc_ofdm_demod(const double* ltf_re, const double* ltf_im, long n_items, int nt, double* out_re, double* out_im)
{
    int bins[OC_NDATA];
    double cs[OC_FFT], sn[OC_FFT];
    oc_data_bins(bins);
    for (int m = 0; m < OC_FFT; ++m) {               /* exact table of e^{-2 pi i m / 256}, quadrant by quadrant */
        const double a = 2.0 * 3.14159265358979323846 * (double)(m % 64) / 256.0;
        const double c = cos(a), s = sin(a);
        switch (m / 64) {
        case 0:  cs[m] =  c; sn[m] = -s; break;
        case 1:  cs[m] = -s; sn[m] = -c; break;
        case 2:  cs[m] = -c; sn[m] =  s; break;
        default: cs[m] =  s; sn[m] =  c; break;
        }
    }
    for (long it = 0; it < n_items; ++it)
        for (int s = 0; s < nt; ++s) {
            const double* xr = ltf_re + ((size_t)it * nt + s) * OC_SYM + OC_CP;
            const double* xi = ltf_im + ((size_t)it * nt + s) * OC_SYM + OC_CP;
            for (int b = 0; b < OC_NDATA; ++b) {
                const int f = (bins[b] - 1 + OC_FFT / 2) % OC_FFT;      /* shifted index i holds DFT bin (i + 128) mod 256 */
                double ar = 0.0, ai = 0.0;
                for (int n = 0; n < OC_FFT; ++n) {
                    const int m = (f * n) % OC_FFT;
                    ar += xr[n] * cs[m] - xi[n] * sn[m];
                    ai += xr[n] * sn[m] + xi[n] * cs[m];
                }
                out_re[((size_t)it * OC_NDATA + b) * nt + s] = ar;
                out_im[((size_t)it * OC_NDATA + b) * nt + s] = ai;
            }
        }
}

/* a-2.  rxsym [n_items][234][nt] -> H [n_items][nt][234] (tx-major rows, the DNN's output layout, BER_test_maMIMO_LTF.m:191-195).
   P [nt][nt], row j = pilot sequence of tx antenna j (MATLAB P(j,:)).  hD(:,j) = rxsym * Puse(:,j) ./ denom with Puse = P'
   (conjugate transpose, helperMIMOChannelEstimate.m:24) and denom = nltf .* ltf(CarriersLocations) (:26-27, :33-36). */
void oc_ls_from_rxsym(const double* rx_re, const double* rx_im, long n_items, int nt, const double* P_re, const double* P_im,
                      double* h_re, double* h_im)
{
    int bins[OC_NDATA];
    double seq[OC_FFT];
    oc_data_bins(bins);
    oc_vht_ltf_256(seq);
    for (long it = 0; it < n_items; ++it)
        for (int j = 0; j < nt; ++j)
            for (int b = 0; b < OC_NDATA; ++b) {
                const double* rr = rx_re + ((size_t)it * OC_NDATA + b) * nt;
                const double* ri = rx_im + ((size_t)it * OC_NDATA + b) * nt;
                double ar = 0.0, ai = 0.0;
                for (int s = 0; s < nt; ++s) {                           /* Puse(s, j) = conj(P(j, s)) */
                    const double pr = P_re[(size_t)j * nt + s], pi = P_im ? -P_im[(size_t)j * nt + s] : 0.0;
                    ar += rr[s] * pr - ri[s] * pi;
                    ai += rr[s] * pi + ri[s] * pr;
                }
                const double denom = (double)nt * seq[bins[b] - 1];
                h_re[((size_t)it * nt + j) * OC_NDATA + b] = ar / denom;
                h_im[((size_t)it * nt + j) * OC_NDATA + b] = ai / denom;
            }
}

/* a-1 + a-2: the whole LS path of n_items preambles.  Returns 0, or -1 when out of memory. */
int oc_ls_estimate(const double* ltf_re, const double* ltf_im, long n_items, int nt, const double* P_re, const double* P_im,
                   double* h_re, double* h_im)
{
    const size_t n = (size_t)n_items * OC_NDATA * nt;
    double* rr = (double*)malloc(n * sizeof(double));
    double* ri = (double*)malloc(n * sizeof(double));
    if (!rr || !ri) { free(rr); free(ri); return -1; }
    oc_ofdm_demod(ltf_re, ltf_im, n_items, nt, rr, ri);
    oc_ls_from_rxsym(rr, ri, n_items, nt, P_re, P_im, h_re, h_im);
    free(rr); free(ri);
    return 0;
}

/* a-4 .. a-7: one component model (real or imag) on rows x [n_rows][d_in] that are already flattened and concatenated
   ([flatten(seq_in), seq_p], massiveMIMO_CSI_prediction_DNN.py:207-208).
   n_hidden hidden blocks: Dense + relu (:211-214), then BatchNormalization on the relu's output (:215-219; keras inference form
   y = (x - moving_mean) / sqrt(moving_variance + eps) * gamma + beta, eps = 1e-3 unless told otherwise), Dropout = identity at
   inference (:222); then Dense linear (:227).
   kernels[i] [in][out] row-major (the keras layout), biases[i] [out]; bn[4 i + 0..3] = gamma, beta, moving_mean, moving_variance
   of block i, or bn == NULL for a model without BatchNormalization.  y [n_rows][n_out].  Returns 0, or -1 when out of memory. */
int oc_fc_forward(const double* x, long n_rows, int d_in, int n_hidden, const int* widths, const double* const* kernels,
                  const double* const* biases, const double* const* bn, double bn_eps, const double* w_reg, const double* b_reg,
                  int n_out, double* y)
{
    enum { RB = 8 };                     /* rows taken together, so that a kernel row is read once per RB samples; every
                                            output is still one plain sum over k in ascending order */
    int wmax = d_in > n_out ? d_in : n_out;
    for (int i = 0; i < n_hidden; ++i) if (widths[i] > wmax) wmax = widths[i];
    double* a = (double*)malloc((size_t)RB * wmax * sizeof(double));
    double* b = (double*)malloc((size_t)RB * wmax * sizeof(double));
    if (!a || !b) { free(a); free(b); return -1; }
    for (long r0 = 0; r0 < n_rows; r0 += RB) {
        const int nr = n_rows - r0 < RB ? (int)(n_rows - r0) : RB;
        int din = d_in;
        for (int r = 0; r < nr; ++r) memcpy(a + (size_t)r * wmax, x + (size_t)(r0 + r) * d_in, (size_t)d_in * sizeof(double));
        for (int l = 0; l <= n_hidden; ++l) {                             /* l == n_hidden: the linear regressor (:227) */
            const int last = l == n_hidden;
            const int dout = last ? n_out : widths[l];
            const double* W = last ? w_reg : kernels[l];
            const double* bias = last ? b_reg : biases[l];
            for (int r = 0; r < nr; ++r) for (int o = 0; o < dout; ++o) b[(size_t)r * wmax + o] = 0.0;
            for (int k = 0; k < din; ++k) {                               /* row k of the kernel is contiguous */
                const double* wk = W + (size_t)k * dout;
                for (int r = 0; r < nr; ++r) {
                    const double ak = a[(size_t)r * wmax + k];
                    double* br = b + (size_t)r * wmax;
                    for (int o = 0; o < dout; ++o) br[o] += ak * wk[o];
                }
            }
            for (int r = 0; r < nr; ++r)
                for (int o = 0; o < dout; ++o) {
                    double v = b[(size_t)r * wmax + o] + bias[o];
                    if (!last) {
                        v = v > 0.0 ? v : 0.0;                            /* relu (:211-214), THEN BatchNormalization (:215-219) */
                        if (bn) v = (v - bn[4 * l + 2][o]) / sqrt(bn[4 * l + 3][o] + bn_eps) * bn[4 * l + 0][o] + bn[4 * l + 1][o];
                    }
                    b[(size_t)r * wmax + o] = v;
                }
            double* t = a; a = b; b = t;
            din = dout;
        }
        for (int r = 0; r < nr; ++r) memcpy(y + (size_t)(r0 + r) * n_out, a + (size_t)r * wmax, (size_t)n_out * sizeof(double));
    }
    free(a); free(b);
    return 0;
}

/* a-3: the network's input rows of a batch of packets, one component (part = the real or the imaginary plane of the preambles
   [n_pkt][nr][len_ltf]).  Row s = p Nr Nt + r Nt + t (create_massiveMIMO_CSIest_dnn_dataset.py:62) = [ part(p, r, :) ; P(t, :) ]
   (massiveMIMO_dataGenerator.py:307-311, massiveMIMO_CSI_prediction_DNN.py:207-208).  x [n_pkt nr nt][len_ltf + nt]. */
void oc_samples_from_packets(const double* part, long n_pkt, int nr, int nt, int len_ltf, const double* P, double* x)
{
    const int d = len_ltf + nt;
    for (long p = 0; p < n_pkt; ++p)
        for (int r = 0; r < nr; ++r)
            for (int t = 0; t < nt; ++t) {
                double* row = x + (((size_t)p * nr + r) * nt + t) * d;
                memcpy(row, part + ((size_t)p * nr + r) * len_ltf, (size_t)len_ltf * sizeof(double));
                memcpy(row + len_ltf, P + (size_t)t * nt, (size_t)nt * sizeof(double));
            }
}

/* f-3: LMMSE_ce.m:23-39, one link.  h_tilde [np] (the link's LS estimate, re / im), h [n_h] the vector the reference passes as
   "channel impulse response" (generate_maMIMO_LTF.m:342 hands it the scatterer delays), snr_db scalar.  out [nfft].
   Rpp x = H_tilde is solved by Gaussian elimination with partial pivoting instead of forming inv(Rpp) (:39) - the same product
   up to the conditioning of Rpp.  Returns 0, -1 when out of memory, -2 for a singular Rpp. */
#include <complex.h>
int oc_lmmse_ce(const double* ht_re, const double* ht_im, int nfft, int np, int nps, const double* h_re, const double* h_im, int n_h,
                double snr_db, double* out_re, double* out_im)
{
    const double snr = pow(10.0, snr_db * 0.1);                                    /* :23 */
    double hh = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < n_h; ++k) {                                                 /* :27-29, k = 0 .. length(h) - 1 */
        const double m2 = h_re[k] * h_re[k] + (h_im ? h_im[k] * h_im[k] : 0.0);
        hh += m2; s1 += m2 * k; s2 += m2 * (double)k * k;
    }
    const double r = s1 / hh, r2 = s2 / hh;
    const double tau_rms = sqrt(r2 - r * r);                                        /* :30 */
    const double df = 1.0 / nfft;                                                   /* :31 */
    const double complex j2 = I * (2.0 * 3.14159265358979323846 * tau_rms * df);    /* :32 */
    double complex* A = (double complex*)malloc((size_t)np * np * sizeof(double complex));
    double complex* x = (double complex*)malloc((size_t)np * sizeof(double complex));
    if (!A || !x) { free(A); free(x); return -1; }
    for (int a = 0; a < np; ++a) {
        for (int b = 0; b < np; ++b)                                                /* :35-36, :38 */
            A[(size_t)a * np + b] = 1.0 / (1.0 + j2 * (double)(nps * (a - b))) + (a == b ? 1.0 / snr : 0.0);
        x[a] = ht_re[a] + I * ht_im[a];
    }
    for (int c = 0; c < np; ++c) {
        int piv = c;
        for (int a = c + 1; a < np; ++a) if (cabs(A[(size_t)a * np + c]) > cabs(A[(size_t)piv * np + c])) piv = a;
        if (cabs(A[(size_t)piv * np + c]) == 0.0) { free(A); free(x); return -2; }
        if (piv != c) {
            for (int b = 0; b < np; ++b) { double complex t = A[(size_t)c * np + b]; A[(size_t)c * np + b] = A[(size_t)piv * np + b]; A[(size_t)piv * np + b] = t; }
            double complex t = x[c]; x[c] = x[piv]; x[piv] = t;
        }
        for (int a = c + 1; a < np; ++a) {
            const double complex f = A[(size_t)a * np + c] / A[(size_t)c * np + c];
            for (int b = c; b < np; ++b) A[(size_t)a * np + b] -= f * A[(size_t)c * np + b];
            x[a] -= f * x[c];
        }
    }
    for (int a = np - 1; a >= 0; --a) {
        double complex s = x[a];
        for (int b = a + 1; b < np; ++b) s -= A[(size_t)a * np + b] * x[b];
        x[a] = s / A[(size_t)a * np + a];
    }
    for (int k1 = 0; k1 < nfft; ++k1) {                                             /* :33-34, :39 */
        double complex s = 0.0;
        for (int k2 = 0; k2 < np; ++k2) s += x[k2] / (1.0 + j2 * (double)(k1 - k2 * nps));
        out_re[k1] = creal(s); out_im[k1] = cimag(s);
    }
    free(A); free(x);
    return 0;
}

/* a-12: NMSE_subk, BER_test_maMIMO_LTF.m:675-686 - per link ||ref - est||^2 / ||ref||^2 over the 234 bins, mean over the links */
double oc_nmse_subk(const double* ref_re, const double* ref_im, const double* est_re, const double* est_im, long n_links, int n_bins)
{
    double acc = 0.0;
    for (long l = 0; l < n_links; ++l) {
        double num = 0.0, den = 0.0;
        for (int k = 0; k < n_bins; ++k) {
            const size_t i = (size_t)l * n_bins + k;
            const double dr = ref_re[i] - est_re[i], di = ref_im[i] - est_im[i];
            num += dr * dr + di * di;
            den += ref_re[i] * ref_re[i] + ref_im[i] * ref_im[i];
        }
        acc += num / den;
    }
    return acc / (double)n_links;
}
