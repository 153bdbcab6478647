/*
 * misonet.h -- C ABI of libmisonet_hip.so: the MI355X (gfx950) MISO1 -> MVDR -> MISO3 inference path.
 *
 * The reference (yuhogun0908/MISOnet) is pure Python and has no FFI layer; its boundary for this
 * path is the Python call surface named in SURVEY.md 8(b).  Each entry point below states the
 * reference interface it replaces (file:line into the reference tree).  All pointers named *_dev are
 * device pointers owned by the caller (e.g. torch tensors); the library allocates device memory only
 * in *_commit (weights).  Every call is asynchronous on the given HIP stream unless stated.  Return
 * value: 0 = success, negative = error (misonet_strerror / misonet_last_error).  One handle per
 * device; a handle is not re-entrant.
 *
 * Spectrogram layout at this boundary is the reference's: complex64 (interleaved re,im float32),
 * [B, channels, T frames, F = 129 bins], F innermost (model.py:77-80, tester.py:1025).
 */
#ifndef MISONET_H_
#define MISONET_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct misonet_net misonet_net;
typedef struct misonet_pipeline misonet_pipeline;
typedef void* misonet_stream;           /* hipStream_t */

enum {
  MISONET_OK = 0,
  MISONET_EINVAL = -1,        /* bad argument / unsupported geometry */
  MISONET_ESTATE = -2,        /* call order (e.g. forward before commit, missing tensor) */
  MISONET_EHIP = -3,          /* a HIP runtime call failed */
  MISONET_ENOMEM = -4,        /* workspace too small */
  MISONET_ENAN = -5           /* NaN in a network output (the reference drops into pdb, model.py:109-110) */
};

/* Geometry of one MISO trunk.  Mirrors the constructor arguments of MISO_1 / MISO_3
 * (model.py:9, 283): in_ch = 2*num_ch (MISO_1, model.py:16) or 2*(num_ch+2) (MISO_3, model.py:290);
 * out_ch = 2*num_spks (model.py:17); en_ch / de_ch = en/de_bottleneck_channels (config/NN_BSS.yml:120-123). */
typedef struct {
  int in_ch;
  int out_ch;
  int en_ch[7];
  int de_ch[7];
  int n_freq;                 /* must be 129 (nperseg 256): the encoder must reduce F to one bin */
  int tcn_norm;               /* ABI 400: `norm_type` of the constructors = the two OUTER norms of every TemporalBlock
                               * (model.py:530,535, chose_norm model.py:570-581): 0 "IN" (nn.InstanceNorm1d, no parameters;
                               * config/NN_BSS.yml:123), 1 "gLN" (GlobalLayerNorm: gamma, beta), 2 "cLN"
                               * (ChannelwiseLayerNorm: gamma, beta), 3 anything else (nn.BatchNorm1d in eval mode: weight,
                               * bias, running_mean, running_var).  The 2-D blocks hard-code InstanceNorm2d (model.py:413,430)
                               * and the norm inside DepthwiseSeparableConv is always gLN (model.py:533,537). */
} misonet_cfg;

const char* misonet_strerror(int code);
const char* misonet_last_error(void);   /* thread-local detail of the last failing call */
int misonet_version(void);

/* ---- network handle: replaces nn.Module construction + load_state_dict (run.py:121-151) --------------- */
int misonet_net_create(const misonet_cfg* cfg, misonet_net** out);
int misonet_net_destroy(misonet_net* net);
/* the state_dict this network expects: the reference's key names (268 tensors for the default cfg) */
int misonet_net_num_tensors(const misonet_net* net);
const char* misonet_net_tensor_name(const misonet_net* net, int i);
long long misonet_net_tensor_numel(const misonet_net* net, int i);
/* host float32 data of one state_dict entry (load_state_dict, run.py:139-151) */
int misonet_net_set_tensor(misonet_net* net, const char* key, const float* host_data, long long numel);
/* all tensors set -> repack into the kernel layouts and upload to the current device (synchronous) */
int misonet_net_commit(misonet_net* net);

/* arithmetic of the 3x3 convolutions (99.4 % of the FLOPs; the reference computes them in float32, model.py:77-80,
 * 401-482):
 *   0 "f32"     exact float32 matrix cores (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain);
 *   3 "bf16x6"  (the DEFAULT of a new handle, and what bench.py reports) fp32-FAITHFUL on the bf16 matrix cores: both operands are represented exactly as three bf16 pieces
 *               (24 bits), the six leading partial products are accumulated in float32 (the dropped ones are < 2^-23
 *               of a product, one float32 rounding); activations travel pre-split (oct3 layout: hi | mid | lo, 8 channels
 *               per 16-byte unit), the instance norm is folded into per-sample weights, staging is LDS-DMA.  Same error
 *               against the reference as mode 0 (2.3e-6 per forward) at 1.65 x its speed;
 *   5 "f32w"    (ABI 430) mode 0 with the DenseBlock convs (model.py:437-482: 94 % of the MACs) in Winograd F(2x2, 3x3) form:
 *               float32 products and sums on the same matrix cores, 2.25 x fewer of them (conv_wino.hip); measured 2.0e-6
 *               per forward against the reference (mode 0: 2.6e-6), 1.47 x mode 0's speed.  Same planar float32 layout.
 * Modes 1, 2 ("bf16x3": 16-bit operands), 4 ("f16x3": 22-bit operands) and 6 ("bf16x6w", ABI 440: the Winograd form in mode 3's
 * arithmetic -- correct, 0.8 x mode 3's speed) are measured alternatives that are NOT product modes (ABI 450): only the experiment
 * build of the library (csrc: `make exp`) contains them, the product library answers MISONET_EINVAL.
 * The choice is internal to the workspace (whose size depends on it: misonet_net_workspace_bytes must be asked again
 * after a change): inputs, outputs and taps are the same float32 / complex64 tensors in every mode. */
int misonet_net_set_precision(misonet_net* net, int mode);
int misonet_net_get_precision(const misonet_net* net);

/* workspace (device bytes) for n_samples spectrograms of n_frames frames */
long long misonet_net_workspace_bytes(const misonet_net* net, int n_samples, int n_frames);

/* MISO_1.forward(mixture) (model.py:76-111) and MISO_3.forward(mixture, a, b) (model.py:350-395).
 * The input is given as n_seg channel segments, each complex64 [B, seg_ch[i], T, F]; their real parts are
 * concatenated in order, then their imaginary parts (model.py:80 / 360-364).  MISO_1: one segment (mixture);
 * MISO_3: three (mixture, a, b).  out_dev: complex64 [B, out_ch/2, T, F]. */
int misonet_net_forward(misonet_net* net, int n_seg, const void* const* seg_dev, const int* seg_ch,
                        int B, int T, void* out_dev, void* ws_dev, long long ws_bytes, misonet_stream stream);
/* synchronises the stream and reports MISONET_ENAN if the last forward on this workspace produced a NaN */
int misonet_net_check(misonet_net* net, const void* ws_dev, misonet_stream stream);

/* Workspace liveness: by default activation buffers whose lifetimes do not overlap share memory (the only cross-level
 * lifetime of the reference is the skip list xs, model.py:84-99): 0.26 GB per forward-sample at T = 1001 in mode 3
 * instead of 0.55.  keep != 0 gives every buffer its own memory so that every tap below stays readable after a forward;
 * the workspace size changes (ask misonet_net_workspace_bytes again). */
int misonet_net_keep_activations(misonet_net* net, int keep);
/* diagnostic (host only, no GPU): the memory plan of ONE sample's activation block for n_frames frames -- per buffer its
 * byte offset inside the block, its size and the first / last step of a forward at which it is alive (0 input written,
 * 1 + b encoder b, 8 TCN, 9 + i decoder i, 16 results read).  Returns the number of buffers written (<= max_buffers) or
 * a negative error.  tests/test_lib_abi.py checks that buffers alive at the same step never overlap. */
int misonet_net_buffer_plan(const misonet_net* net, int n_frames, int max_buffers, long long* offset_bytes,
                            long long* size_bytes, int* first_step, int* last_step);
/* test/diagnostic taps: copy an intermediate activation of the LAST forward (still in ws_dev) out as float32
 * [B, C, T, F] in the reference's layout and normalisation.  Names: enc0_conv, enc0..enc6, tcn_out, dec0..dec6.
 * Needs misonet_net_keep_activations(net, 1) BEFORE that forward (MISONET_ESTATE otherwise), except dec6 (the output). */
int misonet_net_tap_shape(const misonet_net* net, const char* name, int* C, int* F);
int misonet_net_tap(misonet_net* net, const char* name, const void* ws_dev, int B, int T, float* dst_dev,
                    misonet_stream stream);

/* ---- MVDR: Tester_Enhance.Apply_Beamforming(source_stft, mix_stft, epsi) (tester.py:1071-1136) ---------- */
/* src_dev, mix_dev: complex64 [B, F, M, T] contiguous (the reference passes permuted views, tester.py:921-923);
 * out_dev: complex64 [B, T, F] (tester.py:1134).  M <= 8. */
long long misonet_mvdr_workspace_bytes(int B, int F, int M);
int misonet_mvdr(const void* src_dev, const void* mix_dev, int B, int F, int M, int T, float epsi,
                 void* out_dev, void* ws_dev, long long ws_bytes, misonet_stream stream);
/* diagnostic: after misonet_mvdr, copy steering vectors (after phase correction) and beamformer weights,
 * both complex128 [B,F,M], to device buffers (either may be NULL) */
int misonet_mvdr_debug(const void* ws_dev, int B, int F, int M, void* steer_c128_dev, void* w_c128_dev,
                       misonet_stream stream);

/* ---- PIT speaker alignment (tester.py:1043-1065 and 889-915) ----------------------------------------------- */
/* anchor_dev, cand_dev: complex64 [B, S, T, F]; sel_dev: int32 [B, S] with aligned speaker i = cand[sel[i]];
 * dist_dev (required: it is the call's only scratch, so the call allocates nothing and stays asynchronous): float64,
 * B*S*S*(F+1) elements.  The first [B, S, S] receive dist[i][j] = sum_{t,f} | |anchor_i| - |cand_j| |; the rest holds
 * the per-bin partial sums [B, F, S, S], which are added in bin order (no atomics: the distances and the selected
 * permutation are bit-reproducible from run to run).  All S! permutations are enumerated in
 * itertools.permutations order with the first minimum winning, as the reference's einsum('bij,pij->bp') + argmin
 * (tester.py:1053-1064); 1 <= S <= 4.
 * dist_bytes = the size of dist_dev in bytes: less than misonet_pit_scratch_bytes(B, S, F) returns MISONET_ENOMEM (ABI 400;
 * until ABI 300 the size was implicit, and it had grown from B*S*S doubles in ABI 200 without the signature changing). */
long long misonet_pit_scratch_bytes(int B, int S, int F);
int misonet_pit_select(const void* anchor_dev, const void* cand_dev, int B, int S, int T, int F,
                       int* sel_dev, double* dist_dev, long long dist_bytes, misonet_stream stream);

/* ---- fused on-device pipeline: the body of Tester_Enhance.inference (tester.py:865-939) -------------------- */
/* MISO1_Inference (6 circular shifts batched as 6B forwards, tester.py:1014-1068) -> clean-reference
 * alignment (tester.py:889-915; skipped when clean_dev == NULL) -> MVDR per speaker (tester.py:917-924) ->
 * MISO3 per speaker (tester.py:936-939).  Everything stays in HBM in the kernels' own layout. */
/* 2 <= num_mic <= 8; 1 <= num_spk <= 4 (the reference's own clean-reference alignment stacks exactly s0 and s1,
 * tester.py:889-891, i.e. its harness is 2-speaker; here clean_dev simply carries num_spk sources).
 * miso3 == NULL creates a SEPARATION-ONLY pipeline (the body the reference's Tester_Beamforming shares with
 * Tester_Enhance, tester.py:340-449: MISO1_Inference + alignments, no MISO3 memory): misonet_pipeline_run then needs
 * out_dev == NULL and bf_dev == NULL (MISONET_ESTATE otherwise). */
int misonet_pipeline_create(misonet_net* miso1, misonet_net* miso3, int num_mic, int num_spk, int ref_ch,
                            float epsi, misonet_pipeline** out);
int misonet_pipeline_destroy(misonet_pipeline* p);
long long misonet_pipeline_workspace_bytes(const misonet_pipeline* p, int B, int T);
/* mix_dev complex64 [B,M,T,F]; clean_dev complex64 [B,S,T,F] or NULL; out_dev complex64 [B,S,T,F] (MISO3);
 * optional outputs (may be NULL): bf_dev complex64 [B,S,T,F] (MVDR), miso1_dev complex64 [B,S,M,T,F] (aligned).
 * out_dev == NULL with miso1_dev != NULL runs the separation stage only (MISO1_Inference + alignments): the input of
 * the utterance-wise beamformer of Tester_Beamforming (tester.py:340-449). */
int misonet_pipeline_run(misonet_pipeline* p, const void* mix_dev, const void* clean_dev, int B, int T,
                         void* out_dev, void* bf_dev, void* miso1_dev, void* ws_dev, long long ws_bytes,
                         misonet_stream stream);
int misonet_pipeline_check(misonet_pipeline* p, const void* ws_dev, misonet_stream stream);
/* same pipeline fed with waveforms: wav_dev float32 [B, n_samples, M] (time-major, microphones interleaved: the
 * librosa.load(...).T array of dataloader/data.py:605-616), clean_wav_dev float32 [B, n_samples, S] (clean sources at
 * ref_ch) or NULL.  The STFT front-end (data.py:505-522,540-544) runs on the device straight into the kernels' layout;
 * T = n_samples/64 + 1 frames (misonet_pipeline_workspace_bytes takes that T). */
int misonet_pipeline_run_wav(misonet_pipeline* p, const float* wav_dev, const float* clean_wav_dev, int B,
                             int n_samples, void* out_dev, void* bf_dev, void* miso1_dev, void* ws_dev,
                             long long ws_bytes, misonet_stream stream);

/* ---- STFT front-end alone: AudioDataset_Test.STFT + "/scale" + permute (dataloader/data.py:505-522, 540-544) ------ */
/* wav_dev float32 [B, n_samples, M] -> out_dev complex64 [B, M, T, 129], T = n_samples/64 + 1: hann-256, hop 64,
 * zero boundary padding, un-normalised.  The twiddle tables (0.3 MB per device) are built by misonet_net_commit,
 * misonet_pipeline_create or misonet_frontend_init (ABI 430; see misonet_istft below for a process that calls none of them). */
int misonet_stft_frames(int n_samples);
long long misonet_stft_workspace_bytes(int B, int M, int n_samples);
/* STFT + iSTFT tables (0.3 MB) and kernel attributes of the CURRENT device; idempotent (ABI 430) */
int misonet_frontend_init(void);
int misonet_stft(const float* wav_dev, int B, int n_samples, int M, void* out_dev, void* ws_dev, long long ws_bytes,
                 misonet_stream stream);

/* ---- iSTFT + int16: Tester_Enhance.ISTFT and the post-processing of tester.py:949-952, 979-990 ("x scale" ->
 * scipy.signal.istft(hann, 256, 192) -> "x 32767" -> astype(int16)) (ABI 420) ---------------------------------------- */
/* spec_dev complex64 [N, T, 129] (the boundary layout of every output above, T >= 2) -> 64 (T - 1) samples per row:
 * out_i16_dev int16 [N, 64 (T - 1)] (truncating cast of y * 32767) and / or out_f32_dev float32 of the same shape (either
 * may be NULL).  The windowed inverse DFT runs on the fp32 matrix cores, overlap-add and the division by the window
 * envelope follow on chip.  Like every other call it is asynchronous and allocation-free -- PROVIDED the front-end tables of
 * the current device exist: misonet_net_commit, misonet_pipeline_create and misonet_frontend_init build them (hipMalloc +
 * synchronous copy, once per device, thread-safe).  A process that calls misonet_stft / misonet_istft without any of
 * these builds them inside its first call, which therefore synchronises and must not sit inside a stream capture. */
int misonet_istft(const void* spec_dev, int N, int T, void* out_i16_dev, float* out_f32_dev, misonet_stream stream);

/* ---- per-launch timing (bench.py roofline leg): while enabled, the library brackets every conv launch, the TCN
 * section and the MVDR section of each forward with HIP events on the caller's stream.  kinds: 0 = 3x3 conv kernel
 * launches, 1 = TCN sections, 2 = MVDR sections, 3 = other (conv_wprep_k, the per-sample weight preparation of the
 * DMA dataflow).  State is per device (the current device at the call); misonet_profile_end synchronises and sums. */
int misonet_profile_begin(int max_launches);
int misonet_profile_end(double* ms_by_kind /*[4]*/, long long* launches_by_kind /*[4]*/);

/* ---- timing helper: HIP events on the caller's stream (bench.py roofline leg) ------------------------------- */
int misonet_event_create(void** ev);
int misonet_event_record(void* ev, misonet_stream stream);
int misonet_event_elapsed_ms(void* start, void* stop, float* ms);   /* synchronises on stop */
int misonet_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* MISONET_H_ */
