/*
 * df_oracle.h — CPU oracle for the libDF half of DeepFilterNet's enhance() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar C restatement of the reference's Rust DSP core
 * (libDF/src/lib.rs, libDF/src/transforms.rs, pyDF/src/lib.rs).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (deepfilternet_amd) never does.
 *
 * Parity status: the reference crate cannot be built here (no Rust toolchain), and its FFT lives in the
 * un-vendored third-party crates realfft 3.3.0 / rustfft 6.2.0 (libDF/Cargo.toml:97-98).  The FFT below is
 * a restatement of the *published semantics* of those crates (unnormalised R2C / C2R, C2R ignores the
 * imaginary parts of DC and Nyquist), pinned against numpy.fft in tests/test_oracle_dsp.py; absolute STFT bin
 * values are therefore "pinned by numpy identity", not by a reference-authored vector.
 *
 * All arrays are C-contiguous.  Complex numbers are interleaved float pairs (re, im), i.e. Complex32.
 */
#ifndef DF_ORACLE_H
#define DF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dfo_state dfo_state;

/* libDF/src/lib.rs:68-100 erb_fb().  out[nb_bands].  Returns 0, or -1 on bad args. */
int dfo_erb_fb(int sr, int fft_size, int nb_bands, int min_nb_freqs, uint64_t *out);

/* libDF/src/lib.rs:104-154 DFState::new().  Returns NULL if hop*2 > fft (reference: assert/panic). */
dfo_state *dfo_state_new(int sr, int fft_size, int hop_size, int nb_bands, int min_nb_freqs);
void dfo_state_free(dfo_state *st);
/* libDF/src/lib.rs:156-159 reset(): zero analysis/synthesis memories. */
void dfo_state_reset(dfo_state *st);
int dfo_state_sr(const dfo_state *st);
int dfo_state_fft_size(const dfo_state *st);
int dfo_state_hop_size(const dfo_state *st);
int dfo_state_nb_erb(const dfo_state *st);
float dfo_state_wnorm(const dfo_state *st);
void dfo_state_window(const dfo_state *st, float *out /*[fft]*/);
void dfo_state_erb_widths(const dfo_state *st, uint64_t *out /*[nb_erb]*/);

/* libDF/src/lib.rs:356-394 frame_analysis(): one hop of input -> one spectrum frame (scaled by wnorm). */
void dfo_frame_analysis(dfo_state *st, const float *in /*[hop]*/, float *out /*[F][2]*/);
/* libDF/src/lib.rs:396-427 frame_synthesis(): one spectrum frame -> one hop of output. Does not modify `in`. */
void dfo_frame_synthesis(dfo_state *st, const float *in /*[F][2]*/, float *out /*[hop]*/);

/* pyDF/src/lib.rs:41-72 DF.analysis(): x[C][T] -> spec[C][T/hop][F][2]; state reset per channel if reset. */
void dfo_analysis(dfo_state *st, const float *x, int64_t C, int64_t T, int reset, float *spec);
/* pyDF/src/lib.rs:74-107 DF.synthesis(): spec[C][Tf][F][2] -> out[C][Tf*hop]. */
void dfo_synthesis(dfo_state *st, const float *spec, int64_t C, int64_t Tf, int reset, float *out);

/* libDF/src/transforms.rs:236-253 + lib.rs:280-295: spec[rows][F][2] -> out[rows][nb]; dB if db. */
void dfo_erb(const float *spec, int64_t rows, const uint64_t *widths, int nb, int db, float *out);
/* libDF/src/transforms.rs:285-299 + lib.rs:339-348: gains[rows][nb] -> out[rows][F]. */
void dfo_erb_inv(const float *gains, int64_t rows, const uint64_t *widths, int nb, float *out);
/* libDF/src/transforms.rs:301-330 + lib.rs:244-251.  x[C][T][E] in place; state[C][E] or NULL (linspace -60..-90). */
void dfo_erb_norm(float *x, int64_t C, int64_t T, int E, float alpha, float *state);
/* libDF/src/transforms.rs:332-361 + lib.rs:253-259.  x[C][T][F][2] in place; state[C][F] or NULL (linspace 1e-3..1e-4). */
void dfo_unit_norm(float *x, int64_t C, int64_t T, int F, float alpha, float *state);
/* libDF/src/lib.rs:314-326 apply_interp_band_gain() over rows: spec[rows][F][2] *= gains[rows][nb] per band. */
void dfo_apply_band_gain(float *spec, int64_t rows, const float *gains, const uint64_t *widths, int nb);
/* libDF/src/lib.rs:446-471 post_filter() over rows of F bins (F rounded down to a multiple of 4, as chunks_exact). */
void dfo_post_filter(const float *noisy, float *enh, int64_t rows, int F, float beta);
/* pyDF/src/lib.rs:300-309: linspace(1e-3, 1e-4, n). */
void dfo_unit_norm_init(int n, float *out);

#ifdef __cplusplus
}
#endif
#endif
