/* ctcdecode_amd.h -- C ABI of the MI355X-native CTC prefix beam-search decoder.
 *
 * Drop-in boundary for the ONE hot path of parlance/ctcdecode: CTCBeamDecoder.decode() without a language model.
 * Each entry point names the reference interface it replaces (paths relative to the reference checkout):
 *
 *   ctcd_beam_decode        <- ctcdecode/src/binding.cpp:103-120  paddle_beam_decode()  (pybind11: binding.cpp:291)
 *                              = binding.cpp:35-101 beam_decode() -> ctc_beam_search_decoder_batch()
 *                                (ctcdecode/src/ctc_beam_search_decoder.cpp:245-285) with ext_scorer == nullptr
 *   ctcd_beam_decode_host   <- the same call as the reference makes it: CPU tensors in, CPU tensors out
 *                              (ctcdecode/__init__.py:77-123 moves probs to the CPU first; here they are staged to HBM)
 *   ctcd_create / destroy   <- no reference counterpart (the reference allocates per call); caches HBM scratch
 *   ctcd_last_error         <- replaces LOG(FATAL)/abort (ctcdecode/src/decoder_utils.h:17-29) by error codes
 *
 * Plain pointers and sizes only; no torch / HIP types.  `stream` is a hipStream_t passed as void* (NULL = default).
 * All functions return CTCD_OK (0) or a negative CTCD_E* code; ctcd_last_error() describes the last failure of the
 * calling thread.
 *
 * Tensor layouts (row-major, identical to the reference's tensors, ctcdecode/__init__.py:83-86):
 *   probs        float32 [B, T, V]   log_input == 1: log-probabilities; 0: probabilities (the reference's two modes,
 *                                    ctcdecode/__init__.py:42 log_probs_input); 2 (extension): raw logits, normalised on the
 *                                    device by a float32 log_softmax first (see ctcd_log_softmax)
 *   seq_lens     int32   [B] or NULL (all T); each clamped to [0, T]         (binding.cpp:64-65)
 *   out_tokens   int32   [B, beam, T] label ids   of beam p of item b at [b][p][0 .. out_lens[b][p])
 *   out_timesteps int32  [B, beam, T] frame index at which each label's probability peaked (path_trie.cpp:42-45)
 *   out_scores   float32 [B, beam]   negative log-likelihood, ascending = best first (decoder_utils.cpp:68)
 *   out_lens     int32   [B, beam]
 *   n_results    int32   [B] or NULL: number of beams actually returned for item b (min(beam, #prefixes))
 * Everything the reference leaves uninitialised (rows p >= n_results, positions >= out_lens) is written as 0.
 */
#ifndef CTCDECODE_AMD_H_
#define CTCDECODE_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTCD_OK 0
#define CTCD_EINVAL (-1)      /* bad argument (sizes, blank_id, NULL pointers, ...) */
#define CTCD_EUNSUPPORTED (-2) /* configuration outside what this build implements (see ctcd_last_error) */
#define CTCD_EHIP (-3)        /* HIP runtime error */
#define CTCD_EINTERNAL (-4)   /* decoder status word reported a failure */

typedef struct ctcd_decoder ctcd_decoder;

/* Create a decoder bound to HIP device `device_id`.  Owns only scratch (node pools, pruned candidate lists). */
int ctcd_create(ctcd_decoder **out, int device_id);
void ctcd_destroy(ctcd_decoder *dec);

/* Decode a batch whose tensors all live in the HBM of the decoder's device.  Asynchronous on `stream` -- call
 * hipStreamSynchronize (or ctcd_check_status / ctcd_fetch_status_async) before reading.  Nothing is read back or decided on
 * the host in between: the pre-passes (vocabulary pruning when cutoff_top_n < V or cutoff_prob < 1, with its std::sort replay
 * of the frames the fast pass flags; the probability -> log conversion when log_input == 0, binary64 log restated for the
 * device) and the decode kernel are queued behind each other on `stream`; see DESIGN.md 5. */
int ctcd_beam_decode(ctcd_decoder *dec, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                     int num_processes /* accepted for signature parity; unused on the GPU */, double cutoff_prob,
                     int cutoff_top_n, int blank_id, int log_input, int32_t *out_tokens, int32_t *out_timesteps,
                     float *out_scores, int32_t *out_lens, int32_t *n_results, void *stream);

/* log_softmax over the last axis of float32 [B, T, V] device memory, `out` may alias `logits` (no reference counterpart:
 * its callers run torch's log_softmax before decode(), README.md:30-38).  This is what log_input == 2 applies.  Defined
 * to the bit: y_j = (x_j - m) - logf(s), m = max x (NaNs skipped, a zero maximum taken as +0), s = sum_j expf(x_j - m) in float32 with lane l = j mod 64 adding its
 * terms in increasing j and the 64 partial sums combined by a butterfly (^1, ^2, ... ^32); expf / logf as in glibc, expf
 * below -88 taken as 0.  Frames at or beyond seq_lens[b] are not touched.  Asynchronous on `stream`. */
int ctcd_log_softmax(ctcd_decoder *dec, const float *logits, const int32_t *seq_lens, int B, int T, int V, float *out, void *stream);

/* Same, with HOST pointers for every tensor (what paddle_beam_decode receives).  Synchronous: when it returns -- with or
 * without an error -- nothing queued by the call still reads `probs` or writes the outputs.  Both PCIe legs overlap the
 * kernel: log-probability input without pruning is streamed to the already running kernel on a second
 * stream the decoder owns; finished utterances write their results into page-locked host memory themselves and host
 * threads (max(num_processes, 16)) expand them into the padded tensors.  `probs` may be pageable. */
int ctcd_beam_decode_host(ctcd_decoder *dec, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                          int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                          int32_t *out_tokens, int32_t *out_timesteps, float *out_scores, int32_t *out_lens,
                          int32_t *n_results);

/* ---- Compact result delivery (no reference counterpart: the reference fills two padded [B, beam, T] tensors,
 * binding.cpp:85-99).  The beam is a trie, so its label sequences overlap almost entirely; in compact form every beam
 * entry hands over only the labels it does not share with its predecessor in trie (DFS) order:
 *   c_hdr    int32 [B][4]        {#results, #labels of the item, index of its first label in c_labels, 0}
 *   c_ent    int32 [B][beam][4]  per entry j in DFS order: {result row, #labels shared with entry j-1, length, index of its own labels}
 *   c_labels uint32 [capacity]   label | frame << 16 (T <= 65536, V <= 65535), one bump-allocated buffer for the batch
 *   c_count  uint32 [1]          labels used (device word, zeroed by the call)
 * ctcd_beam_decode_compact = ctcd_beam_decode[_lm] (scorer may be NULL) writing this form (everything device memory);
 * ctcd_expand_compact rebuilds out_tokens / out_timesteps [B, beam, T] on the device (e.g. on the rank that gathered them);
 * ctcd_beam_decode_to_host is the reference's whole call -- results as HOST tensors, inputs on either side -- with the
 * compact form crossing PCIe and `num_processes` host threads expanding it. */
long long ctcd_compact_label_capacity(int B, int beam, int T);
typedef struct ctcd_scorer ctcd_scorer;
int ctcd_beam_decode_compact(ctcd_decoder *dec, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                             int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                             ctcd_scorer *scorer, int32_t *c_hdr, int32_t *c_ent, uint32_t *c_labels, uint32_t *c_count,
                             long long label_capacity, float *out_scores, int32_t *out_lens, int32_t *n_results, void *stream);
int ctcd_expand_compact(ctcd_decoder *dec, const int32_t *c_hdr, const int32_t *c_ent, const uint32_t *c_labels, int B, int beam,
                        int T, int32_t *out_tokens, int32_t *out_timesteps, void *stream);
int ctcd_beam_decode_to_host(ctcd_decoder *dec, const float *probs, const int32_t *seq_lens, int probs_on_device, int B, int T, int V,
                             int beam, int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                             ctcd_scorer *scorer, int32_t *out_tokens, int32_t *out_timesteps, float *out_scores,
                             int32_t *out_lens, int32_t *n_results, void *stream);

/* ---- LM tier: the external scorer.  Replaces ctcdecode/src/binding.cpp:122-150,263-287:
 *   ctcd_scorer_create          <- paddle_get_scorer(alpha, beta, lm_path, labels, vocab_size)   (binding.cpp:143-150)
 *   ctcd_scorer_destroy         <- paddle_release_scorer                                          (binding.cpp:263-265)
 *   ctcd_scorer_is_character_based / _max_order / _dict_size / _reset_params <- binding.cpp:271-287
 *   ctcd_beam_decode_lm[_host]  <- paddle_beam_decode_lm: paddle_beam_decode plus `void *scorer`  (binding.cpp:122-140)
 * The scorer object (ctcdecode/src/scorer.h:41-110) is built on the host from an ARPA text model (binary kenlm files are
 * not supported) and the label strings; its tables -- n-gram back-off automaton, dictionary trie -- are mirrored into the
 * HBM of `device_id` once, and the per-frame queries of DecoderState::next() run inside the decode kernel.  `labels`:
 * V NUL-terminated UTF-8 strings.  Word models need a " " label and at most 64 labels.  LM arithmetic follows kenlm's
 * published algorithm (float32 weights, longest listed n-gram, back-off weights added in float32); see DESIGN.md for
 * what this parity is pinned to. */
int ctcd_scorer_create(ctcd_scorer **out, double alpha, double beta, const char *lm_path, const char *const *labels, int V,
                       int device_id);
void ctcd_scorer_destroy(ctcd_scorer *scorer);
int ctcd_scorer_is_character_based(const ctcd_scorer *scorer);
int ctcd_scorer_max_order(const ctcd_scorer *scorer);
int ctcd_scorer_dict_size(const ctcd_scorer *scorer);
int ctcd_scorer_reset_params(ctcd_scorer *scorer, double alpha, double beta);
/* Scorer::get_log_cond_prob (scorer.cpp:74-93) on explicit words, evaluated on the host copy of the tables (tests). */
double ctcd_scorer_cond_log_prob(const ctcd_scorer *scorer, const char *const *words, int n);
/* ---- The swappable scorer: a scorer whose language model lives behind a HOST callback.  The reference's decoder takes an
 * opaque `void *scorer` (binding.cpp:122-140) and only ever calls Scorer::get_log_cond_prob(words) on the language-model side
 * (scorer.h:41-78, scorer.cpp:74-93; called from ctc_beam_search_decoder.cpp:120-137 with the window make_ngram built,
 * scorer.cpp:163-194); this is that interface across the C ABI, for models the built-in ARPA tables do not cover (binary
 * kenlm files through the `kenlm` module, a neural LM, a remote service ...).
 *   fn(user, words, n, &log10_prob): the window's n = max_order words, oldest first, "<s>"-padded as make_ngram pads them;
 *     store log10 p(words[n-1] | words[0..n-2]) as float32 -- what kenlm's BaseScore returns, the decoder applies the
 *     reference's conversion p / NUM_FLT_LOGE (scorer.cpp:92) -- and return 0; return 1 if the window holds a word the model
 *     does not know (the reference's OOV_SCORE, scorer.cpp:86-88); return < 0 to fail the decode.  The value must be
 *     finite (NaN or +-inf with return code 0 fails the decode: -inf is how the cache marks an OOV answer).  Must be a pure function
 *     of the words: answers are cached on the device ((history, word) -> log10 prob, the same tables the built-in scorer
 *     queries inside the kernel) and each distinct window is asked for once per scorer.  Called on the thread that calls
 *     the decode, between kernel launches: a launch runs until an utterance needs a window that is not cached, parks that
 *     utterance at the frame boundary, and resumes after the host has asked.
 *   vocabulary: the model's words (what Scorer::fill_dictionary reads from the model, scorer.cpp:196-230): builds the
 *     dictionary of a word model; all entries single characters <=> character model (scorer.cpp:65-71).
 *   Results equal those of a scorer that knew every answer from the start (tests/test_gpu_lm.py: the built-in tables behind
 *   the callback give bit-identical output).  Accepted by ctcd_beam_decode_lm, ctcd_beam_decode_lm_host,
 *   ctcd_beam_decode_to_host, ctcd_beam_decode_compact (since round 5: the compact multi-GPU gather works with it) and
 *   ctcd_stream_create_lm / ctcd_stream_decode.
 *   Since round 6 without shape restrictions: rows that hold +-inf or overflow float32 sums, and beams of any width the decoder
 *   takes (the wide-beam layouts), decode behind a callback as they do with the built-in tables.
 *   Decodes that share one callback scorer are serialised (its cache is one object) and the callback runs under that lock:
 *   a callback that itself decodes with the same scorer deadlocks.  A ctcd_stream_decode call that fails half-way (the
 *   callback reported an error) leaves the streams of that call unusable: destroy them.
 *   Cost: a warm cache decodes in one launch, like the built-in tables.  Cold, an utterance misses once per frame in which a
 *   prefix completes a word under a history it has not asked about; since round 6 its workgroup WAITS for the answer (see
 *   ctcd_last_scorer_waits below) instead of ending its launch, and the call is bound by the calling thread: the callback's own
 *   time plus ~0.12 us of bookkeeping per queued pair (bench.py "scorer hook": 128 x 1500 frames of transcript-like rows under a
 *   5-gram model, 195 k windows, 86-88 ms cold in one launch -- 264 ms and 699 launches in round 5).  ctcd_last_scorer_rounds tells
 *   how many launches the last call took.  The cache of such a scorer starts with room for 2 M windows (64 + 32 MB of HBM).
 * ctcd_scorer_cond_log10 evaluates any scorer in the callback's own form (so the built-in tables can sit behind one);
 * ctcd_scorer_callback_calls counts the callback invocations so far (= distinct windows cached). */
typedef int (*ctcd_cond_log10_fn)(void *user, const char *const *words, int n, float *log10_prob);
int ctcd_scorer_create_callback(ctcd_scorer **out, double alpha, double beta, int max_order, const char *const *vocabulary,
                                int n_vocabulary, ctcd_cond_log10_fn fn, void *user, const char *const *labels, int V, int device_id);
int ctcd_scorer_cond_log10(const ctcd_scorer *scorer, const char *const *words, int n, float *log10_prob);
long long ctcd_scorer_callback_calls(const ctcd_scorer *scorer);
/* seconds spent inside the callback so far (wall time of the asking passes of waiting launches; elsewhere every 16th call is timed
 * and counted sixteen times) */
double ctcd_scorer_callback_seconds(const ctcd_scorer *scorer);
/* A callback that may be called from several threads at once (a read-only model behind it; NOT a Python callable: the interpreter
 * lock serialises those): `threads` - 1 helper threads ask beside the calling thread while a launch waits for its answers -- the new
 * windows of a batch of queued pairs are split among them; caching the answers stays with the calling thread, so results and the
 * "every window is asked once" property are unchanged.  The helpers spin while a waiting launch is served and sleep otherwise.
 * 1 (the default): the callback is only ever called from the thread that called the decoder.  threads in [1, 64]. */
int ctcd_scorer_set_callback_threads(ctcd_scorer *scorer, int threads);
/* launches the decoder's last call through a callback scorer took (1 = the cache held everything; one more per round of
 * misses): what a cold / lukewarm cache costs (bench.py "scorer hook") */
int ctcd_last_scorer_rounds(ctcd_decoder *dec);
/* Since round 6 a launch WAITS for the callback's answers: a workgroup whose utterance misses stays on its CU, the calling thread
 * -- polling a miss list in page-locked memory -- asks the callback and publishes the answered cache slots, and the workgroup takes
 * the utterance up again; a relaunch remains for what cannot be done under a running kernel (the cache table or the state arrays
 * have to grow).  ctcd_last_scorer_waits: the answer batches the last call's launches were handed that way;
 * ctcd_set_scorer_wait(dec, 0): every miss ends the utterance's launch, as in rounds 4-5 (identical results; tests run both). */
int ctcd_last_scorer_waits(ctcd_decoder *dec);
int ctcd_set_scorer_wait(ctcd_decoder *dec, int on);

int ctcd_beam_decode_lm(ctcd_decoder *dec, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                        int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, ctcd_scorer *scorer,
                        int32_t *out_tokens, int32_t *out_timesteps, float *out_scores, int32_t *out_lens, int32_t *n_results,
                        void *stream);
int ctcd_beam_decode_lm_host(ctcd_decoder *dec, const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam,
                             int num_processes, double cutoff_prob, int cutoff_top_n, int blank_id, int log_input,
                             ctcd_scorer *scorer, int32_t *out_tokens, int32_t *out_timesteps, float *out_scores,
                             int32_t *out_lens, int32_t *n_results);

/* ---- Streaming ("online") decoding: replaces ctcdecode/src/binding.cpp:153-265 (paddle_get_decoder_state,
 * paddle_beam_decode_with_given_state, paddle_release_state) = DecoderState objects kept alive between calls
 * (ctcdecode/src/ctc_beam_search_decoder.cpp:230-243,288-317).  A ctcd_stream parks one utterance's beam and node pool
 * in HBM between launches; frame numbers in `timesteps` keep counting across chunks (abs_time_step, :69).
 * ctcd_stream_decode feeds chunk b (probs[b, 0:min(seq_lens_host[b], T)], DEVICE pointer) to states[b]; where
 * is_eos[b] != 0 the stream's final beams are written to row b of the outputs (row stride out_T >= total frames of that
 * stream), elsewhere row b stays zero and n_results[b] = 0.  seq_lens_host and is_eos are HOST arrays (B entries). */
typedef struct ctcd_stream ctcd_stream;
int ctcd_stream_create(ctcd_decoder *dec, ctcd_stream **out, int V, int beam, int frames_hint);
/* paddle_get_decoder_state with a scorer (binding.cpp:243-261): the stream decodes with the LM tier */
int ctcd_stream_create_lm(ctcd_decoder *dec, ctcd_stream **out, int V, int beam, int frames_hint, ctcd_scorer *scorer);
void ctcd_stream_destroy(ctcd_decoder *dec, ctcd_stream *st);
long long ctcd_stream_frames(const ctcd_stream *st);
int ctcd_stream_decode(ctcd_decoder *dec, ctcd_stream **states, const unsigned char *is_eos, const float *probs,
                       const int32_t *seq_lens_host, int B, int T, int V, int beam, int num_processes, double cutoff_prob,
                       int cutoff_top_n, int blank_id, int log_input, int32_t *out_tokens, int32_t *out_timesteps,
                       float *out_scores, int32_t *out_lens, int32_t *n_results, int out_T, void *stream);

/* The same call with the results delivered to HOST memory in the reference's own sizes (binding.cpp:186-205 resizes its tensors
 * to [B, R, L]: R = the most results of any stream that ended, L = the longest of their label sequences).  R and L exist only
 * once the kernel has run, so the caller passes an allocator: it is called once, with (R, L), and returns the two int32 buffers
 * of B * R * L elements (return non-zero to fail the call; with R * L == 0 the buffers are not touched).  The finished streams'
 * results cross PCIe in compact form and host threads expand them (as ctcd_beam_decode_to_host does for the one-shot call).
 * `probs` is a DEVICE pointer; out_scores / out_lens [B, beam], n_results [B], *out_R, *out_L are HOST memory.  Synchronous.
 * Streams behind a callback scorer end through ctcd_stream_decode (CTCD_EUNSUPPORTED here). */
typedef int (*ctcd_result_alloc_fn)(void *user, int R, int L, int32_t **tokens, int32_t **timesteps);
int ctcd_stream_decode_to_host(ctcd_decoder *dec, ctcd_stream **states, const unsigned char *is_eos, const float *probs,
                               const int32_t *seq_lens_host, int B, int T, int V, int beam, int num_processes, double cutoff_prob,
                               int cutoff_top_n, int blank_id, int log_input, ctcd_result_alloc_fn alloc, void *alloc_user,
                               float *out_scores, int32_t *out_lens, int32_t *n_results, int out_T, int *out_R, int *out_L, void *stream);

/* Host check of the per-item status words written by the last ctcd_beam_decode (synchronises the device). */
int ctcd_check_status(ctcd_decoder *dec, int B);
/* ... without blocking: enqueues the copy of the B status words (0 = ok) into `host_status` (page-locked memory) on
 * `stream` -- the launch stream, before anything else is enqueued there; the caller synchronises on its own event.  For
 * pipelined callers that queue the next batch before looking at this one. */
int ctcd_fetch_status_async(ctcd_decoder *dec, int B, int32_t *host_status, void *stream);

/* Vocabulary prune bookkeeping of the last ctcd_beam_decode (0 for the no-prune configurations).  A frame in which equal
 * values sit at the cutoff_top_n boundary or among the kept ones is ordered by whatever std::sort does with the whole
 * row (decoder_utils.cpp:19-20), and a cumulative sum next to cutoff_prob depends on the reference's sequential chain of
 * binary64 log / exp (decoder_utils.cpp:25-32): those frames are "flagged" by the fast prune pass and settled by a second
 * device kernel that replays libstdc++'s std::sort and walks the chain with bit-exact restatements of glibc's log / exp
 * (exact_math_f64.h).  Nothing reaches the host toolchain since round 4: ctcd_last_prune_host_rows is always 0.
 * ctcd_last_prune_flagged_rows waits for the call's stream (the count arrives behind the kernels). */
long long ctcd_last_prune_flagged_rows(ctcd_decoder *dec);
long long ctcd_last_prune_host_rows(ctcd_decoder *dec);

/* HIP-event timing of the decode kernel alone (events recorded on the launch stream). */
int ctcd_set_timing(ctcd_decoder *dec, int on);
int ctcd_last_kernel_ms(ctcd_decoder *dec, float *ms);
int ctcd_last_prune_ms(ctcd_decoder *dec, float *ms); /* the vocabulary-prune kernel of the same call (pruned configurations) */

/* Test hook: device expf/logf/log_sum_exp (exact_math.h; modes 0..2) and binary64 log / exp / log_sum_exp<double>
 * (exact_math_f64.h; modes 3: log(p), 4: log(p + FLT_MIN), 5: exp(x) over float bit patterns, 6: pairs) vs the host C library. */
int ctcd_debug_math_check(ctcd_decoder *dec, int mode, uint32_t lo, uint32_t hi, uint32_t stride, const float *xs,
                          const float *ys, long long n_pairs, long long *checked, long long *mismatches);

/* Test/tuning hook: instrumented build of the kernel; out = int64 [B][16] per-phase timer ticks. */
int ctcd_debug_set_profile(ctcd_decoder *dec, int on);
int ctcd_debug_get_profile(ctcd_decoder *dec, long long *out, int B);
/* Test hook: 1 (default) = beams <= 128 over <= 32 labels run the kernel variant with a compile-time workspace layout,
 * 0 = always the run-time layout (identical results). */
int ctcd_debug_set_fixed_layout(ctcd_decoder *dec, int on);
/* Test hook: 1 (default) = the std::sort replay of flagged prune frames works in LDS when the row fits, 0 = always in
 * global memory, the path of rows beyond ~11 000 labels (identical results). */
int ctcd_debug_set_prune_resolve(ctcd_decoder *dec, int on);
/* Test hook for log_input == 2: 1 (default) = rows of more than 256 labels (a multiple of four) are normalised by a workgroup
 * each, and in front of a vocabulary prune not at all -- one kernel reads the logits and emits the kept candidates' normalised
 * values; 0 = always the one-wave log_softmax pass followed by the separate prune (identical results). */
int ctcd_debug_set_fused_logits(ctcd_decoder *dec, int on);
/* Test hook: the workgroup form of the vocabulary-prune pass (rows of 257 .. 10 240 labels, a multiple of four, cutoff_top_n <= 64) keeps
 * the row in registers between its two looks at it (1, default) or reads it twice (0: the form of rounds 2-5).  Identical results. */
int ctcd_debug_set_prune_registers(ctcd_decoder *dec, int on);
/* Test hook: the vocabulary-prune pass's output of the last call (get_pruned_log_probs, decoder_utils.cpp:10-45, per frame), copied
 * to HOST memory: cnt[rows], labels / values [rows][stride], stride = min(cutoff_top_n, V); entries at or beyond a frame's count
 * are unspecified.  Waits for the launch stream. */
int ctcd_debug_prune_rows(ctcd_decoder *dec, long long rows, int stride, int32_t *cnt, int32_t *labels, float *values);
/* Tuning aid (instrumented build): per-wave shader-clock stamps at every workgroup barrier of batch item 0 during frames
 * [frame0, frame0 + nframes).  out == NULL arms the following decodes; out != NULL (int64 [16][ctcd_debug_timeline_cap()])
 * fetches the stamps, arrival and departure alternating, in program order (tools/barrier_timeline.py). */
int ctcd_debug_timeline_cap(void);
int ctcd_debug_timeline(ctcd_decoder *dec, int frame0, int nframes, long long *out);
/* Debug aid (instrumented build): beam of batch item 0 after every frame, int32 [T][1 + 4*beam] = n, then
 * (node, depth, lcp, score bits) per entry.  Call with on=1 before a decode, then with out != NULL to fetch. */
int ctcd_debug_beam_dump(ctcd_decoder *dec, int on, int *out, int T, int beam);

/* Test hook for the host-tensor entry points (ctcd_beam_decode_to_host / _host / _lm_host): input_streaming = 0 / 1 turns
 * the streamed input off / on (-1: leave; on by default: the kernel is launched before its rows have crossed PCIe and
 * waits for them frame block by frame block; 2: on, but the "frames arrived" counter is never advanced -- the kernel must
 * give up after about a second and the call repeat itself the plain way); mirror_cap_labels >= 0 shrinks the page-locked mirror a finished utterance
 * copies its compact results into (utterances beyond it are fetched from the device buffer afterwards; -1: default size,
 * -2: leave). */
int ctcd_debug_set_host_path(ctcd_decoder *dec, int input_streaming, long long mirror_cap_labels);

/* Tuning / introspection. */
int ctcd_set_threads(ctcd_decoder *dec, int threads_per_workgroup); /* 0 = automatic (default); else a power of two in [64, 1024] */
/* Two workgroups per CU.  The fixed-layout class (beam <= 128, <= 32 labels) has a second build of its kernel that fits
 * two utterances on a compute unit (<= 64 VGPRs; the exact replay's scratch in HBM: 67 KB of LDS instead of 132 KB): a lone
 * workgroup runs ~7 % slower in it, two on a CU together 1.4-1.5x faster.  mode = -1 (default): used when a batch has
 * more utterances than the device has CUs; 1: always (a serving loop that keeps several launches in flight); 0: never. */
int ctcd_set_cu_sharing(ctcd_decoder *dec, int mode);
/* The north-star class of shapes (beam <= 128, <= 32 labels, no scorer) has a second build of its kernel whose phase A1 settles
 * four subtrees per wavefront at once: beams shaped like chains -- what blank-dominated rows, i.e. real acoustic posteriors,
 * produce: some twenty entries with descendants in the beam per frame -- decode 7 % faster with it, bushy beams (random rows:
 * one or two) 3 % slower.  mode -1 (default): chosen per launch from the shape statistic of the last launch whose status was
 * checked (ctcd_check_status; the *_host / to_host entries check); 0 / 1: never / always.  Results are identical in both
 * builds.  ctcd_last_subtree_search: which build the last launch used (0 / 1). */
int ctcd_set_subtree_search(ctcd_decoder *dec, int mode);
int ctcd_last_subtree_search(const ctcd_decoder *dec);
int ctcd_workgroup_lds_bytes(int beam, int V, int cutoff_top_n, double cutoff_prob); /* LDS one utterance needs (default build) */
const char *ctcd_last_error(void);
const char *ctcd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CTCDECODE_AMD_H_ */
