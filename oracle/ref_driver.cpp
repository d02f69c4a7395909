// oracle/_ref driver -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Links the reference's own, unmodified no-LM decoder sources
// (/root/reference/ctcdecode/src/{ctc_beam_search_decoder,path_trie,decoder_utils}.cpp,
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

// Link stubs: the decoder translation unit references these Scorer members from
// branches that are unreachable when ext_scorer == nullptr
// (ctc_beam_search_decoder.cpp:133-134,179-180,201,205).
std::vector<std::string> Scorer::make_ngram(PathTrie *) { abort(); }
double Scorer::get_log_cond_prob(const std::vector<std::string> &) { abort(); }
double Scorer::get_sent_log_prob(const std::vector<std::string> &) { abort(); }
std::vector<std::string> Scorer::split_labels(const std::vector<int> &) { abort(); }

extern "C" int ctcref_decode_f32(const float *probs, const int32_t *seq_lens, int B, int T, int V,
                                 int beam, int num_processes, double cutoff_prob, int cutoff_top_n,
                                 int blank_id, int log_input, int32_t *out_tokens,
                                 int32_t *out_timesteps, float *out_scores, int32_t *out_lens,
                                 int32_t *n_results) {
  std::vector<std::string> vocab(V);
  for (int i = 0; i < V; ++i) vocab[i] = "#" + std::to_string(i);  // no " ": space_id = -2, unused without scorer
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
                                           log_input, nullptr);
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
