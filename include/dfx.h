/*
 * dfx.h — C ABI of libdfx.so, the MI355X (gfx950) engine for DeepFilterNet's enhance() hot path.
 *
 * Drop-in boundary.  Every entry point names the reference interface it replaces
 * (paths relative to the reference checkout):
 *   - pyDF/src/lib.rs (pyo3 module `libdf`)  : class DF + erb / erb_inv / erb_norm / unit_norm / unit_norm_init
 *   - libDF/src/lib.rs, libDF/src/transforms.rs : DFState and the batched transforms behind it
 *   - DeepFilterNet/df/multiframe.py:160-180, DeepFilterNet/df/modules.py:226-269 : MF.DF and Mask (device ops)
 *   - DeepFilterNet/df/deepfilternet3.py:334-456, DeepFilterNet/df/enhance.py:190-250 : DfNet.forward, enhance()
 *   - libDF/src/capi.rs:83-253 : naming / ownership conventions (opaque handles, caller-owned buffers)
 *
 * Conventions
 *   - plain C types only; complex numbers are interleaved float pairs (re, im) == Complex32 == numpy complex64
 *   - every data pointer is a DEVICE pointer on the current HIP device unless the name ends in `_host`
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on that stream
 *   - all arrays are dense row-major ("C-contiguous") unless a stride argument is given
 *   - return value: DFX_OK or an error code; dfx_last_error() gives a thread-local message.  Nothing panics/aborts.
 *   - no hidden host<->device copies in the batched entry points; nothing here falls back to a CPU implementation
 */
#ifndef DFX_H
#define DFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFX_OK 0
#define DFX_ERR_INVALID_ARG 1
#define DFX_ERR_UNSUPPORTED 2
#define DFX_ERR_HIP 3
#define DFX_ERR_NO_DEVICE 4
#define DFX_ERR_ALLOC 5

#define DFX_VERSION 200 /* 0.2.0 */

int dfx_version(void);
const char *dfx_status_string(int status);
const char *dfx_last_error(void);
/* Number of usable HIP devices (0 => every compute entry point returns DFX_ERR_NO_DEVICE). */
int dfx_device_count(void);
/* 1 when this library is the CPU SIMT interpreter build used by unit tests (tests/hipemu), 0 for the product. */
int dfx_is_emulator(void);

/* ------------------------------------------------------------------------------------------------------------------
 * ERB band table.  Replaces the `erb_fb: &[usize]` argument of libDF (lib.rs:280-348) / pyDF (lib.rs:142-250).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct dfx_bands dfx_bands;
int dfx_bands_create(const uint64_t *widths_host, int nb_bands, dfx_bands **out);
void dfx_bands_free(dfx_bands *b);
int dfx_bands_nb(const dfx_bands *b);
int dfx_bands_nfreqs(const dfx_bands *b);

/* ------------------------------------------------------------------------------------------------------------------
 * DF state.  Replaces DFState::new (libDF/src/lib.rs:104-154) and pyDF's class DF (pyDF/src/lib.rs:14-136).
 * The handle is immutable after creation (window, twiddles, band table live on the device); streaming memories are
 * explicit caller-owned buffers (mem_in / mem_out below), so one handle serves any number of concurrent batches.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct dfx_state dfx_state;
/* Fails with DFX_ERR_INVALID_ARG if hop*2 > fft (lib.rs:111 assert) and DFX_ERR_UNSUPPORTED if fft/2 has a prime
 * factor other than 2, 3, 5 or fft > 4096. */
int dfx_state_create(int sr, int fft_size, int hop_size, int nb_bands, int min_nb_erb_freqs, dfx_state **out);
void dfx_state_free(dfx_state *st);
int dfx_state_sr(const dfx_state *st);
int dfx_state_fft_size(const dfx_state *st);
int dfx_state_hop_size(const dfx_state *st);
int dfx_state_nb_erb(const dfx_state *st);
float dfx_state_wnorm(const dfx_state *st);
const dfx_bands *dfx_state_bands(const dfx_state *st);
int dfx_state_erb_widths(const dfx_state *st, uint64_t *out_host /*[nb_erb]*/);   /* DF.erb_widths() */
int dfx_state_fft_window(const dfx_state *st, float *out_host /*[fft_size]*/);    /* DF.fft_window() */
/* libDF/src/lib.rs:68-100 erb_fb() on the host (pure index arithmetic, bit-exact). */
int dfx_erb_fb(int sr, int fft_size, int nb_bands, int min_nb_freqs, uint64_t *out_host);

/* DF.analysis (pyDF/src/lib.rs:41-72 -> lib.rs:356-394), batched over B independent rows.
 *   x      [B, T] f32            (row stride x_stride elements, >= T)
 *   spec   [B, T/hop, F][2] f32  (trailing T%hop samples are dropped, like chunks_exact)
 *   mem_in [B, fft-hop] or NULL  analysis memory before the first frame (NULL == reset == zeros)
 *   mem_out[B, fft-hop] or NULL  analysis memory after the last frame */
int dfx_analysis(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride, const float *mem_in,
                 float *mem_out, float *spec, void *stream);

/* DF.synthesis (pyDF/src/lib.rs:74-107 -> lib.rs:396-427), batched.  Does NOT modify `spec` (the reference clobbers
 * it).  imag(DC) and imag(Nyquist) are ignored (lib.rs:398-405).
 *   spec [B, Tf, F][2];  out [B, Tf*hop] (row stride out_stride);  mem_* [B, fft-hop] synthesis overlap memory. */
int dfx_synthesis(const dfx_state *st, const float *spec, int64_t B, int64_t Tf, const float *mem_in, float *mem_out,
                  float *out, int64_t out_stride, void *stream);

/* erb() (pyDF/src/lib.rs:142-192 -> transforms.rs:236-253, lib.rs:280-295): spec [rows, F][2] -> out [rows, nb]. */
int dfx_erb(const dfx_bands *bands, const float *spec, int64_t rows, int db, float *out, void *stream);
/* erb_inv() (pyDF/src/lib.rs:194-250 -> lib.rs:339-348): gains [rows, nb] -> out [rows, F]. */
int dfx_erb_inv(const dfx_bands *bands, const float *gains, int64_t rows, float *out, void *stream);
/* erb_norm() (pyDF/src/lib.rs:252-274 -> transforms.rs:301-330, lib.rs:244-251).  x [C, T, E] normalised IN PLACE.
 * state [C, E] in/out or NULL (NULL: linspace(-60,-90,E) per row, final state discarded). */
int dfx_erb_norm(float *x, int64_t C, int64_t T, int E, float alpha, float *state, void *stream);
/* unit_norm() (pyDF/src/lib.rs:276-298 -> transforms.rs:332-361, lib.rs:253-259).
 * x [C, T, F][2] read with frame stride x_frame_stride complex elements (lets spec[..., :nb_df] be read without a copy);
 * out [C, T, F][2] dense.  state [C, F] in/out or NULL (NULL: linspace(1e-3,1e-4,F)). */
int dfx_unit_norm(const float *x, int64_t x_frame_stride, float *out, int64_t C, int64_t T, int F, float alpha,
                  float *state, void *stream);
/* unit_norm_init() (pyDF/src/lib.rs:300-309). */
int dfx_unit_norm_init(int n, float *out_host);

/* df_features() (DeepFilterNet/df/enhance.py:190-203) fused on the device: analysis + erb(dB) + erb_norm + unit_norm.
 *   x [B, T] -> spec [B, Tf, F][2], erb_feat [B, Tf, nb_erb], spec_feat [B, Tf, nb_df][2];  states reset per row. */
int dfx_features(const dfx_state *st, const float *x, int64_t B, int64_t T, int64_t x_stride, int nb_df, float alpha,
                 float *spec, float *erb_feat, float *spec_feat, void *stream);

/* Deep filtering + ERB gains + post filter + attenuation limit, one fused memory-bound kernel.
 * Replaces MF.DF.forward (multiframe.py:169-180), Mask.forward (modules.py:248-269; == lib.rs:314-326), the combine
 * step of DfNet.forward (deepfilternet3.py:442-443), its post filter (:448-454; == lib.rs:446-471) and enhance()'s
 * atten_lim mix (enhance.py:238-240).
 *   spec  [B, T, F][2]   noisy spectrum
 *   coefs complex, nb_df bins, `order` taps:  layout DFX_COEF_BOTF = [B, O, T, nb_df][2]  (MF.DF's input)
 *                                             layout DFX_COEF_BTFO = [B, T, nb_df, O][2]  (DfDecoder's raw output)
 *                                             layout DFX_COEF_BTOF = [B, T, O, nb_df][2]
 *   gains [B, T, nb_bands] or NULL.  NULL: bins >= nb_df are copied from spec (plain MF.DF semantics).
 *   out   [B, T, F][2]   must not alias spec
 *   out[b,t,f<nb_df]  = sum_n coefs[b,n,t,f] * spec[b, t+n-(order-1-lookahead), f]   (zero outside [0,T))
 *   out[b,t,f>=nb_df] = spec[b,t,f] * gains[b,t,band(f)]
 *   pf_beta > 0: post filter;  atten_lim in (0,1): out = spec*atten_lim + out*(1-atten_lim). */
#define DFX_COEF_BOTF 0
#define DFX_COEF_BTFO 1
#define DFX_COEF_BTOF 2 /* [B, T, O, nb_df][2] */
int dfx_df_apply(const float *spec, const float *coefs, int coef_layout, const float *gains, const dfx_bands *bands,
                 int64_t B, int64_t T, int F, int nb_df, int order, int lookahead, float pf_beta, float atten_lim,
                 float *out, void *stream);
/* The same operator on rows with a stride (complex elements, >= F): spec [B,T,spec_stride][2] -> out [B,T,out_stride][2].  Even
 * strides make every row 16-byte aligned (F = fft/2 + 1 is odd for every shipped model) and select the row-streaming kernel the
 * engine uses on its own padded buffers; pad bins of `out` are written as zeros.  No reference counterpart (layout extension). */
int dfx_df_apply_strided(const float *spec, int64_t spec_stride, const float *coefs, int coef_layout, const float *gains,
                         const dfx_bands *bands, int64_t B, int64_t T, int F, int nb_df, int order, int lookahead, float pf_beta,
                         float atten_lim, float *out, int64_t out_stride, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * DeepFilterNet3 model.  Replaces df.deepfilternet3.DfNet (deepfilternet3.py:334-456) for inference.
 * ---------------------------------------------------------------------------------------------------------------- */
#define DFX_SKIP_NONE 0
#define DFX_SKIP_IDENTITY 1
#define DFX_SKIP_GROUPEDLINEAR 2

typedef struct dfx_model_cfg {
    /* [df] section (config.py:12-39) */
    int32_t sr, fft_size, hop_size, nb_erb, nb_df, min_nb_freqs, df_order, df_lookahead;
    int32_t lsnr_min, lsnr_max;
    /* [deepfilternet] section (deepfilternet3.py:25-77) */
    int32_t conv_lookahead, conv_ch, emb_hidden_dim, emb_num_layers, df_hidden_dim, df_num_layers;
    int32_t df_gru_skip; /* DFX_SKIP_* */
    int32_t df_pathway_kernel_size_t, lin_groups, enc_lin_groups;
    int32_t mask_pf;
    float pf_beta;
    float norm_alpha; /* utils.py:111-127 get_norm_alpha() */
    /* skip connections around the embedding GRUs (SqueezedGRU_S gru_skip_op, modules.py:702-738: x = linear_out(gru(linear_in(in))) +
     * skip(in); deepfilternet3.py:138-146 encoder, :198-206 ERB decoder) and the encoder's combine op (:132-136): DFX_SKIP_* / 0|1 */
    int32_t emb_gru_skip_enc, emb_gru_skip, enc_concat;
} dfx_model_cfg;

/* Tensor manifest: the tensors of the reference state-dict the engine consumes, in state_dict() order, with their
 * reference names ("enc.erb_conv0.1.weight", ...).  The host packs them (raw, un-folded float32) into one blob. */
int dfx_model_tensor_count(const dfx_model_cfg *cfg, int *count);
int dfx_model_tensor_info(const dfx_model_cfg *cfg, int index, char *name_out, int name_cap, int64_t shape_out[4],
                          int *ndim_out, int64_t *offset_out /* in floats */);
int dfx_model_blob_floats(const dfx_model_cfg *cfg, int64_t *n);

typedef struct dfx_model dfx_model;
/* Folds BatchNorm (eval mode, eps 1e-5) into the preceding convolution, re-lays the weights out for the kernels and
 * uploads them.  blob_host: float32[dfx_model_blob_floats] packed per dfx_model_tensor_info. */
int dfx_model_create(const dfx_model_cfg *cfg, const float *blob_host, dfx_model **out);
void dfx_model_free(dfx_model *m);
/* .dfx model file = the configuration + the packed blob above ("DFXM", version, cfg, float32 data; written by
 * deepfilternet_amd.export_dfx from a reference model directory or state-dict).  It is what df_create() (include/df_capi.h, the
 * reference's C API) takes as its model path, in place of the reference's tar.gz of ONNX graphs (tract.rs:37-70). */
int dfx_model_save_file(const dfx_model_cfg *cfg, const float *blob_host, const char *path);
/* Takes either file kind (told apart by their magic bytes): a .dfx file, or the reference's own `<model>_onnx.tar.gz` (below). */
int dfx_model_load_file(const char *path, dfx_model **out);
/* The reference's shipped model artefact (libDF/src/tract.rs:29-70 DfParams::from_targz; written by
 * DeepFilterNet/df/scripts/export.py:331-337): a gzip'ed tar of enc.onnx, erb_dec.onnx, df_dec.onnx and config.ini.  The DSP
 * parameters come from the config.ini keys DfTract::new reads (tract.rs:264-315), the network structure and the weights from the
 * three ONNX graphs (BatchNorm as folded by the exporter, GRU gates re-ordered z,r,h -> r,z,n); the result is the
 * (configuration, packed state-dict blob) pair of dfx_model_create().  blob_out may be NULL to query *blob_floats. */
int dfx_onnx_targz_read(const char *path, dfx_model_cfg *cfg_out, float *blob_out, int64_t blob_cap, int64_t *blob_floats);
int dfx_model_cfg_get(const dfx_model *m, dfx_model_cfg *out);
/* The forward pass runs its independent branches (ERB encoder/decoder | DF encoder/decoder | df_convp) on internal
 * streams, forked from and joined to the caller's stream with events.  enable = 0 serialises everything on the caller's
 * stream (useful for per-kernel timing).  Default: enabled (environment DFX_STREAMS=0 disables at creation). */
int dfx_model_set_streams(dfx_model *m, int enable);
/* DfNet(run_df=False) (deepfilternet3.py:383,433-443; init_df(mask_only=True), enhance.py:109,172-175): the DF decoder does not run,
 * the enhanced spectrum is the masked spectrum on every bin, df_coefs is not produced (a caller's buffer is zero-filled). */
int dfx_model_set_run_df(dfx_model *m, int enable);
/* Pipelining knobs (defaults 12, 32, 1; time_chunks <= 16; environment DFX_TCHUNKS / DFX_CHUNKS at creation):
 *   time_chunks      the GRU phase is cut into this many time chunks and every GRU layer runs on its own stream, chunk k of
 *                    layer l starting when layer l-1 has produced chunk k (chain = T*(1 + 2/K) steps instead of 3T);
 *   min_chunk_frames shortest chunk worth a launch;
 *   batch_chunks     dfx_enhance additionally pipelines this many batch chunks (multiples of 16 clips) on separate stream sets.
 * The streams need their own hardware queues: export GPU_MAX_HW_QUEUES=24 before HIP initialises (ROCm maps streams onto 4 queues by
 * default; the Python package sets it on import if it is unset).  The persistent, flag-synchronised GRU phase REQUIRES that its ~8
 * streams make progress independently; the engine checks that with a handshake between them (DFX_Q_HWQ_PROBE) — once per handle, before the
 * first pass that would use the persistent form (handles that are only streamed through never pay for it) — and selects the
 * event-synchronised form when they do not.  Co-residency with another process's kernels is not under the engine's control: a
 * starved flag wait then ends as a reported DFX_ERR_HIP (see dfx_model_check), never as silent garbage or a hang. */
int dfx_model_set_pipeline(dfx_model *m, int time_chunks, int min_chunk_frames, int batch_chunks);
/* Faults a kernel can find while it runs — an activation that left the range of the fp16-split matrix kernels (|x| >= 6e4 at a
 * split: DFX_ERR_UNSUPPORTED; DFX_EXACT_FP32=1 selects the exact fp32 kernels), a flag wait of the persistent GRU phase that timed out
 * (bounded spins, the engine never hangs: DFX_ERR_HIP) — are raised in error words
 * of the model that the device writes and the host reads.  The results of the pass that raised one are INVALID, and no call hides
 * that: every entry point that starts work on the model (dfx_enhance, dfx_model_forward, dfx_stream_process[_raw]) first looks at
 * the words and returns the error of the PREVIOUS pass instead of starting (each word is cleared, atomically, by being reported; a big pass
 * waits for its predecessor anyway, see dfx_enhance).  So a fault of a BIG pass (>= 16384 frames) is reported by the next call at the
 * latest; small passes are not waited for, so a fault of small pass N may only have been raised when call N + 2 looks — a hard
 * guarantee for the last results of a sequence takes dfx_model_check (or DFX_CHECK_EVERY_PASS=1);
 *   dfx_model_poll   looks now, without waiting (faults of work that has completed);
 *   dfx_model_check  waits for the device, then looks: call it before trusting the LAST results of a sequence of calls;
 *   DFX_CHECK_EVERY_PASS=1 (environment, read at dfx_model_create): every call waits for its own pass and reports its own faults.
 * The Python surface (enhance(), DfNet.__call__, DfStream.process) and the reference C API (df_process_frame: aborts like the
 * reference's panics) do this themselves. */
int dfx_model_poll(const dfx_model *m);
int dfx_model_check(const dfx_model *m);
/* Read-only facts about a model handle (what = DFX_Q_*). */
#define DFX_Q_GRU_PERSISTENT 1 /* 1: big multi-stream passes run the GRU phase as ONE persistent, flag-synchronised launch (default on the GPU) */
#define DFX_Q_HWQ_PROBE 2      /* 1: the streams of that phase were seen to run concurrently (the handshake runs at the first use or at this query); 0: they were not (hardware
                                * queues shared: GPU_MAX_HW_QUEUES too small or HIP initialised before it was set) and the event-synchronised
                                * form is used instead; -1: not probed (no streams, exact fp32, CPU interpreter) */
#define DFX_Q_EXACT_FP32 3     /* 1: DFX_EXACT_FP32=1 was set at creation */
#define DFX_Q_SPIN_LIMIT 4     /* polls a flag wait makes before it gives up (DFX_SYNC_SPIN_LIMIT, default 2^22 ~ 2 s) */
#define DFX_Q_PASSES_PERSISTENT 5   /* big passes of this handle that ran the persistent GRU phase */
#define DFX_Q_PASSES_TICKET_BUSY 6  /* big passes that took the event-synchronised form because another PROCESS held the device's ticket for its own
                                       persistent phase (/dev/shm/dfx_persistent_<PCI bus id>.lock, DFX_DEVICE_TICKET=0: no ticket) */
int dfx_model_query(const dfx_model *m, int what, int64_t *value);

/* Scratch memory the caller must provide (device bytes) for a [B, T-frames] batch. */
int dfx_model_workspace_bytes(const dfx_model *m, int64_t B, int64_t T, int64_t *bytes);

/* DfNet.forward (deepfilternet3.py:389-456).
 *   spec [B,T,F][2], feat_erb [B,T,E], feat_spec [B,T,nb_df][2]  ->
 *   spec_e [B,T,F][2], mask [B,T,E] (may be NULL), lsnr [B,T] (may be NULL), df_coefs [B,O,T,nb_df][2] (may be NULL;
 *   DFX_COEF_BOTF == the reference's DfOutputReshapeMF layout, deepfilternet3.py:268-275)
 *   atten_lim: enhance()'s mix factor (0 = off), applied after the optional post filter. */
int dfx_model_forward(const dfx_model *m, const dfx_bands *bands, const float *spec, const float *feat_erb,
                      const float *feat_spec, int64_t B, int64_t T, float atten_lim, float *spec_e, float *mask,
                      float *lsnr, float *df_coefs, void *workspace, int64_t workspace_bytes, void *stream);

/* enhance() (enhance.py:206-250) end to end on the device: x [B, T] -> y [B, T] (pad=1) or y [B, (T/hop)*hop] (pad=0,
 * delayed by fft-hop like the reference).  atten_lim_db: 0 = off.  workspace from dfx_enhance_workspace_bytes.
 * Asynchronous: the call returns before the pass has run (y is valid once `stream` has caught up).  For passes of >= 16384 frames the
 * engine paces its own enqueue (DFX_ENQUEUE_AHEAD=1: off): the call first waits, on the host, until the previous pass of this handle
 * (dfx_enhance or dfx_model_forward) has drained, and it returns only when the encoder front of its own pass has run and the rest is
 * enqueued — packets queued ahead on the pass's hardware queues slow the kernels that are running (docs/measurements.md §5e).  One pass at a time
 * per handle, as before; use one handle per concurrent caller. */
int dfx_enhance_workspace_bytes(const dfx_model *m, const dfx_state *st, int64_t B, int64_t T, int pad, int64_t *bytes);
int dfx_enhance(const dfx_model *m, const dfx_state *st, const float *x, int64_t B, int64_t T, int pad,
                float atten_lim_db, float *y, void *workspace, int64_t workspace_bytes, void *stream);
/* The same pass from and to 16-bit PCM, the sample format the reference's file loop moves (df/enhance.py:73-89 -> df/io.py:25-57 load_audio:
 * torchaudio's int16 normalisation x / 32768; df/io.py:60-84 save_audio: (audio * (1 << 15)).to(torch.int16), truncation toward zero).  Both
 * conversions run inside the STFT kernel's loads and the ISTFT kernel's stores: half the bytes at the boundary (and over PCIe), no conversion
 * launches, and bit-identical samples to dfx_pcm16_to_f32 -> dfx_enhance -> dfx_f32_to_pcm16.  Same workspace, same asynchrony. */
int dfx_enhance_pcm16(const dfx_model *m, const dfx_state *st, const int16_t *x, int64_t B, int64_t T, int pad,
                      float atten_lim_db, int16_t *y, void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Streaming: the frame loop of the reference's real-time runtime, libDF/src/tract.rs `DfTract::process` (:509-642), exported by
 * its C API as df_create / df_get_frame_length / df_set_atten_lim / df_set_post_filter_beta / df_process_frame / df_free
 * (libDF/src/capi.rs:83-253), for `streams` independent mono streams advanced in lockstep on one GPU.
 *   dfx_stream_process(x [streams, n*hop]) -> y [streams, n*hop]: n hops per stream and call (1 <= n <= max_frames; n = 1 is
 *   df_process_frame).  All state lives in the handle: STFT/ISTFT memories, the running ERB / unit-norm means, the convolution
 *   input histories, the hidden states of the five GRU layers and the rolling spectra the deep filter reads.
 *   Like the reference the output is delayed by `lookahead` hops (max(conv_lookahead, df_lookahead), dfx_stream_delay_frames)
 *   on top of the fft-hop samples of the STFT: output hop k is the enhanced input hop k - lookahead; the first `lookahead`
 *   output hops are zero (tract.rs: the rolling spectra start as zeros).  Concatenated over calls, the output equals
 *   dfx_enhance(pad=0) of the whole signal delayed by `lookahead` hops — however the signal is cut into calls.
 *   Stage gating (off by default; dfx_stream_set_gating): the reference decides per frame from the encoder's local SNR whether the
 *   ERB decoder (stage 1) and the DF decoder (stage 2) run — `apply_stages`, tract.rs:658-672, thresholds of RuntimeParams
 *   (:160-189, defaults -10 / 30 / 20 dB) — and answers a stream that has been silent (mean square < 1e-7) for more than 5 hops with
 *   zeros and lsnr = -15 without processing it (:513-525).  With gating on every stream takes these decisions on its own, hop by
 *   hop: lsnr < min: zero mask, no DF; lsnr > max_erb: the spectrum passes unchanged; lsnr > max_df: mask only; otherwise mask +
 *   DF.  A decoder that is skipped keeps its state (GRU hidden states, the delay line in front of df_convp), a frozen stream all of
 *   its state, exactly like tract's pulsed sub-models, which only advance when they are run.  A call of n hops is then n passes.
 *   Multi-channel streams (dfx_stream_set_channels; RuntimeParams::n_ch, tract.rs:119-176; df_create itself is mono, capi.rs:83-104):
 *   `channels` consecutive rows are the channels of one stream.  Every channel has its own STFT / feature / network state; the ERB
 *   masks of a stream's channels are reduced to one mask (ReduceMask: 0 none, 1 max, 2 mean = the reference default, :96-118,868-902)
 *   that is applied to all of them; with gating the silent-input test runs over all channels of the hop and the stage decision is
 *   taken from the first channel's local SNR (:468), one decision and one skip counter per stream.
 * lsnr (optional) receives the local SNR estimate [streams, n] in dB (df_process_frame's return value); not meaningful for the
 * warm-up hops.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct dfx_stream_state dfx_stream_state;
int dfx_stream_create(const dfx_model *m, const dfx_state *st, int64_t streams, int max_frames, dfx_stream_state **out);
void dfx_stream_free(dfx_stream_state *s);
int dfx_stream_reset(dfx_stream_state *s, void *stream);                 /* back to the state after create */
int dfx_stream_frame_length(const dfx_stream_state *s);                  /* hop size in samples (df_get_frame_length) */
int dfx_stream_delay_frames(const dfx_stream_state *s);                  /* lookahead in hops */
int dfx_stream_set_atten_lim(dfx_stream_state *s, float lim_db);         /* df_set_atten_lim: |dB| >= 100 off, < 0.01 bypass */
int dfx_stream_set_post_filter_beta(dfx_stream_state *s, float beta);    /* df_set_post_filter_beta: 0 disables the post filter */
int dfx_stream_set_channels(dfx_stream_state *s, int channels, int reduce_mask); /* rows per stream; 0 none / 1 max / 2 mean */
int dfx_stream_set_gating(dfx_stream_state *s, int enable);              /* DfTract::process's per-frame stage decisions, per stream */
int dfx_stream_set_thresholds(dfx_stream_state *s, float min_db_thresh, float max_db_erb_thresh,
                              float max_db_df_thresh);                   /* RuntimeParams::with_thresholds (tract.rs:160-170) */
int dfx_stream_process(dfx_stream_state *s, const float *x, int64_t n_frames, float *y, float *lsnr, void *stream);
/* DfTract::process_raw (tract.rs:441-507) == df_process_frame_raw (capi.rs:172-210) for every stream: one spectral frame
 * spec [streams, F][2] in, the pass's raw ERB gains [streams, nb_erb] and deep-filter coefficients [streams, df_order, nb_df][2] out,
 * stages [streams] saying which of them exist (bit value 2: gains — the network's mask, or zeros below min_db_thresh; 8: coefficients;
 * the reference signals "absent" with NULL pointers).  Gating must be on; no STFT / deep filtering / synthesis happens here. */
int dfx_stream_process_raw(dfx_stream_state *s, const float *spec, float *gains, float *coefs, unsigned char *stages, float *lsnr,
                           void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-frame Wiener / MVDR filters: df.multiframe.MfWf.forward (multiframe.py:282-321) and MfMvdr.forward (:373-413), the filter
 * stage of the reference's DeepFilterNetMF model (deepfilternetmf.py:335-352).  op 0 = MfWf, 1 = MfMvdr; frame_size N <= 8.
 *   spec [B,T,F][2], ifc [B,T,nb,N][2] (speech inter-frame correlation), mat [B,T,nb,N,N][2] (row-major; an inverse correlation
 *   matrix, a correlation matrix, or a Cholesky factor of either: cholesky_decomp / inverse exactly as the module's constructor
 *   flags, enforce_constraints / eps / dload likewise, defaults 1 / 1e-8 / 1e-7)  ->  out [B,T,F][2] (out != spec; bins >= nb are
 *   copied).  Y[t,f] = sum_n w[n] X[t + n - (N-1-lookahead), f], zero outside the clip (MultiFrameModule.pad :72-76).
 * ---------------------------------------------------------------------------------------------------------------- */
int dfx_mf_filter(const float *spec, const float *ifc, const float *mat, int op, int frame_size, int lookahead, int cholesky_decomp,
                  int inverse, int enforce_constraints, float eps, float dload, int64_t B, int64_t T, int F, int nb, float *out,
                  void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Either side of enhance() in the reference's file loop (df/enhance.py:73-89: load_audio -> enhance -> resample back -> save_audio):
 *   dfx_pcm16_to_f32   torchaudio.load's normalisation of 16-bit PCM, x / 32768 (df/io.py:48)
 *   dfx_f32_to_pcm16   save_audio's encoding, (audio * (1 << 15)).to(int16) (df/io.py:79-80): truncation toward zero, wrap-around
 *   dfx_resample       df.io.resample (io.py:114-116) == torchaudio.functional.resample(audio, orig_sr, new_sr, **params): polyphase
 *                      windowed-sinc bank, method 0 = "sinc_interp_hann", 1 = "sinc_interp_kaiser" (beta); io.py:92-111 parameter sets:
 *                      sinc_fast (0, width 16, rolloff 0.99) | sinc_best (0, 64, 0.99) | kaiser_fast (1, 16, 0.85, beta 8.555504641634386)
 *                      | kaiser_best (1, 16, 0.9475937167399596, 14.769656459379492).  x [B, T] (row stride x_stride) ->
 *                      y [B, dfx_resampler_out_len(T)] (row stride y_stride), device pointers.
 *   dfx_resampler_kernel   the filter bank itself (host arithmetic, no device): W [phases][taps] float32.
 * ---------------------------------------------------------------------------------------------------------------- */
int dfx_pcm16_to_f32(const int16_t *pcm, int64_t n, float *out, void *stream);
int dfx_f32_to_pcm16(const float *x, int64_t n, int16_t *out, void *stream);
typedef struct dfx_resampler dfx_resampler;
int dfx_resampler_create(int orig_sr, int new_sr, int lowpass_filter_width, double rolloff, int method, double beta,
                         dfx_resampler **out);
void dfx_resampler_free(dfx_resampler *r);
int64_t dfx_resampler_out_len(const dfx_resampler *r, int64_t in_len);   /* ceil(new_sr * in_len / orig_sr) */
int dfx_resampler_kernel(int orig_sr, int new_sr, int lowpass_filter_width, double rolloff, int method, double beta, int *phases,
                         int *taps, int *width, float *w_host /* may be NULL */, int64_t cap_floats);
int dfx_resample(const dfx_resampler *r, const float *x, int64_t B, int64_t T, int64_t x_stride, float *y, int64_t y_stride,
                 void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Per-kernel timing (measurement aid, no reference counterpart; the reference only logs wall-clock RTF,
 * DeepFilterNet/df/enhance.py:77-87).  When a kernel's bit is set in `kernel_mask`, every launch of it is bracketed by
 * two hipEvents recorded on the stream the kernel is launched on.  dfx_prof_read() synchronises the pending events and
 * returns the accumulated device time and launch count since the last reset.  Mask 0 (default) = no events at all.
 * ---------------------------------------------------------------------------------------------------------------- */
int dfx_prof_kernel_count(void);
const char *dfx_prof_kernel_name(int kernel_id);
int dfx_prof_enable(uint32_t kernel_mask);
int dfx_prof_reset(void);
int dfx_prof_read(int kernel_id, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* DFX_H */
