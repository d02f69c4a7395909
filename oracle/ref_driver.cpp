// oracle/_ref driver -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Links the reference's own, unmodified decoder sources
// (/root/reference/ctcdecode/src/{ctc_beam_search_decoder,path_trie,decoder_utils,scorer}.cpp,
// compiled in place by oracle/Makefile) behind a small C ABI so that Python
// tests and bench.py's cpu_baseline leg can call the *real* reference.
//
// The marshalling below plays the role of ctcdecode/src/binding.cpp:55-99
// (binding.cpp itself needs torch + boost::python and is not built):
//   * float32 [B,T,V] -> vector<vector<vector<double>>>, each item truncated to
//     min(seq_len, T)                                   (binding.cpp:63-74)
//   * ctc_beam_search_decoder_batch(...)                (binding.cpp:77-78)
//   * results scattered into the four caller-owned arrays, only [b][p][0:len)
//     and p < n_results are written                     (binding.cpp:85-99)
// Extra (not in the reference): n_results[b] is reported so tests know which
// rows are defined (the reference leaves the others uninitialised).
#include <cstdint>
#include <string>
#include <vector>

#include "ctc_beam_search_decoder.h"

// Labels travel as one buffer of NUL-terminated UTF-8 strings (V of them); nullptr = anonymous labels without a space.
static std::vector<std::string> unpack_labels(const char *labels, int V) {
  std::vector<std::string> vocab(V);
  for (int i = 0; i < V; ++i) {
    if (labels) {
      vocab[i] = labels;
      labels += vocab[i].size() + 1;
    } else {
      vocab[i] = "#" + std::to_string(i);  // no " ": space_id = -2, unused without scorer
    }
  }
  return vocab;
}

static int run_batch(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int num_processes,
                     double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, const std::vector<std::string> &vocab,
                     Scorer *scorer, int32_t *out_tokens, int32_t *out_timesteps, float *out_scores, int32_t *out_lens,
                     int32_t *n_results);

extern "C" int ctcref_decode_f32(const float *probs, const int32_t *seq_lens, int B, int T, int V,
                                 int beam, int num_processes, double cutoff_prob, int cutoff_top_n,
                                 int blank_id, int log_input, int32_t *out_tokens,
                                 int32_t *out_timesteps, float *out_scores, int32_t *out_lens,
                                 int32_t *n_results) {
  return run_batch(probs, seq_lens, B, T, V, beam, num_processes, cutoff_prob, cutoff_top_n, blank_id, log_input,
                   unpack_labels(nullptr, V), nullptr, out_tokens, out_timesteps, out_scores, out_lens, n_results);
}

// ---- LM tier: the reference's own Scorer (scorer.cpp, compiled unmodified) over the kenlm / OpenFST stand-ins of
// oracle/shim (the third-party sources are absent from the reference tree).  binding.cpp:122-150,263-287.
extern "C" void *ctcref_scorer_create(double alpha, double beta, const char *lm_path, const char *labels, int V) {
  return new Scorer(alpha, beta, lm_path, unpack_labels(labels, V));  // paddle_get_scorer, binding.cpp:143-150
}
extern "C" void ctcref_scorer_release(void *scorer) { delete static_cast<Scorer *>(scorer); }  // binding.cpp:263-265
extern "C" int ctcref_scorer_is_character_based(void *scorer) { return static_cast<Scorer *>(scorer)->is_character_based(); }
extern "C" int ctcref_scorer_max_order(void *scorer) { return (int)static_cast<Scorer *>(scorer)->get_max_order(); }
extern "C" int ctcref_scorer_dict_size(void *scorer) { return (int)static_cast<Scorer *>(scorer)->get_dict_size(); }
extern "C" void ctcref_scorer_reset_params(void *scorer, double alpha, double beta) { static_cast<Scorer *>(scorer)->reset_params(alpha, beta); }
// words: n NUL-terminated strings; get_log_cond_prob (scorer.cpp:74-93) / get_sent_log_prob (:95-109)
extern "C" double ctcref_scorer_cond_logprob(void *scorer, const char *words, int n) {
  return static_cast<Scorer *>(scorer)->get_log_cond_prob(unpack_labels(words, n));
}
extern "C" double ctcref_scorer_sent_logprob(void *scorer, const char *words, int n) {
  return static_cast<Scorer *>(scorer)->get_sent_log_prob(unpack_labels(words, n));
}

// paddle_beam_decode_lm (binding.cpp:122-140)
extern "C" int ctcref_decode_lm_f32(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int num_processes,
                                    double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, const char *labels,
                                    void *scorer, int32_t *out_tokens, int32_t *out_timesteps, float *out_scores,
                                    int32_t *out_lens, int32_t *n_results) {
  return run_batch(probs, seq_lens, B, T, V, beam, num_processes, cutoff_prob, cutoff_top_n, blank_id, log_input,
                   unpack_labels(labels, V), static_cast<Scorer *>(scorer), out_tokens, out_timesteps, out_scores, out_lens, n_results);
}

static int run_batch(const float *probs, const int32_t *seq_lens, int B, int T, int V, int beam, int num_processes,
                     double cutoff_prob, int cutoff_top_n, int blank_id, int log_input, const std::vector<std::string> &vocab,
                     Scorer *scorer, int32_t *out_tokens, int32_t *out_timesteps, float *out_scores, int32_t *out_lens,
                     int32_t *n_results) {
  std::vector<std::vector<std::vector<double>>> inputs;
  inputs.reserve(B);
  for (int b = 0; b < B; ++b) {
    int len = seq_lens ? seq_lens[b] : T;
    if (len > T) len = T;
    if (len < 0) len = 0;
    std::vector<std::vector<double>> item(len, std::vector<double>(V));
    for (int t = 0; t < len; ++t)
      for (int v = 0; v < V; ++v) item[t][v] = probs[((size_t)b * T + t) * V + v];
    inputs.push_back(std::move(item));
  }
  auto res = ctc_beam_search_decoder_batch(inputs, vocab, (size_t)beam, (size_t)num_processes,
                                           cutoff_prob, (size_t)cutoff_top_n, (size_t)blank_id,
                                           log_input, scorer);
  for (int b = 0; b < B; ++b) {
    const auto &r = res[b];
    if (n_results) n_results[b] = (int32_t)r.size();
    for (size_t p = 0; p < r.size(); ++p) {
      const Output &o = r[p].second;
      size_t base = ((size_t)b * beam + p) * T;
      for (size_t t = 0; t < o.tokens.size(); ++t) {
        out_tokens[base + t] = o.tokens[t];
        out_timesteps[base + t] = o.timesteps[t];
      }
      out_scores[(size_t)b * beam + p] = (float)r[p].first;
      out_lens[(size_t)b * beam + p] = (int32_t)o.tokens.size();
    }
  }
  return 1;  // binding.cpp:100
}
