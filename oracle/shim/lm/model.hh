// Stand-in for kenlm's lm/model.hh (kenlm is an un-vendored submodule of the reference, commit unpinned:
// /root/reference/.gitmodules:1-3; the directory third_party/kenlm is empty).
//
// TEST INFRASTRUCTURE ONLY.  It lets the reference's *unmodified* scorer.cpp compile into oracle/_ref/, so that the
// reference's own LM hooks (make_ngram, get_log_cond_prob, get_sent_log_prob, split_labels, the dictionary FST, and
// the LM branches of DecoderState::next()/decode()) are the code that runs; only the third-party query below is a
// restatement.  LM arithmetic parity is therefore pinned to THIS restatement of kenlm, not to kenlm itself.
//
// What is restated (kenlm's published algorithm, lm/read_arpa.cc + lm/model.cc of github.com/kpu/kenlm):
//   * ARPA text model: "\data\", "ngram N=count", "\N-grams:" sections of lines  log10prob <tab> w1 .. wN [<tab> log10backoff],
//     "\end\".  Numbers are rounded to float32 once (kenlm stores float).  <unk> has index 0; the other words are
//     numbered in file order from 1; the vocabulary callback sees <unk> first, then the words in file order.
//     A word of a higher-order n-gram that is not a unigram maps to <unk>.
//   * GenericModel::FullScore: p(w | state) = log10 prob of the LONGEST n-gram (context suffix + w) listed, plus the
//     back-off weights of the state's context suffixes that are longer than the matched one, added in float32 from
//     the shorter context to the longer.  The out-state keeps the matched n-gram's words (at most order-1) and their
//     back-off weights.  (kenlm additionally drops context words that cannot matter -- zero back-off and no
//     extension; that minimisation never changes a score and is not restated.)
//   * NullContextWrite: the empty state.  BaseVocabulary().Index(): exact string lookup, 0 when unknown.
// Binary kenlm models are not supported by this stand-in.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "lm/config.hh"
#include "lm/state.hh"
#include "lm/virtual_interface.hh"

namespace lm {
namespace ngram {

class ShimVocabulary : public base::Vocabulary {
 public:
  WordIndex Index(const StringPiece &str) const override {
    auto it = ids_.find(std::string(str.data(), str.length()));
    return it == ids_.end() ? 0 : it->second;
  }
  WordIndex Insert(const std::string &w, EnumerateVocab *cb) {
    if (w == "<unk>") return 0;
    auto it = ids_.find(w);
    if (it != ids_.end()) return it->second;
    const WordIndex id = (WordIndex)ids_.size() + 1;
    ids_.emplace(w, id);
    if (cb) cb->Add(id, StringPiece(w));
    return id;
  }

 private:
  std::unordered_map<std::string, WordIndex> ids_;
};

class ShimModel : public base::Model {
 public:
  struct Entry {
    float prob, backoff;
  };
  ShimModel(const char *path, const Config &config) {
    std::ifstream in(path);
    if (!in) die("cannot open", path);
    if (config.enumerate_vocab) config.enumerate_vocab->Add(0, StringPiece("<unk>", 5));
    std::string line;
    std::vector<size_t> counts;
    bool seen_data = false;
    int section = 0;
    uni_.push_back(Entry{-100.0f, -0.0f});  // <unk> when the file has none (kenlm: unknown_missing_logprob)
    while (std::getline(in, line)) {
      while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
      if (line.empty()) continue;
      if (line == "\\data\\") { seen_data = true; continue; }
      if (!seen_data) continue;
      if (line.compare(0, 6, "ngram ") == 0) {
        const size_t eq = line.find('=');
        counts.push_back((size_t)std::strtoull(line.c_str() + eq + 1, nullptr, 10));
        continue;
      }
      if (line == "\\end\\") break;
      if (line[0] == '\\') {  // "\N-grams:"
        section = std::atoi(line.c_str() + 1);
        if (section < 1 || section > (int)counts.size()) die("bad section header in", path);
        continue;
      }
      if (section == 0) die("n-gram line before a section header in", path);
      std::vector<std::string> tok;
      {
        std::istringstream ss(line);
        std::string t;
        while (ss >> t) tok.push_back(t);
      }
      if ((int)tok.size() < 1 + section || (int)tok.size() > 2 + section) die("malformed n-gram line in", path);
      Entry e;
      e.prob = std::strtof(tok[0].c_str(), nullptr);
      if (e.prob > 0.0f) die("positive log probability in", path);
      e.backoff = (int)tok.size() == 2 + section ? std::strtof(tok[1 + section].c_str(), nullptr) : -0.0f;
      if (e.backoff == 0.0f) e.backoff = -0.0f;  // lm/read_arpa.hh ReadBackoff: "always make zero negative"
      if (section == 1) {
        const WordIndex id = vocab_.Insert(tok[1], config.enumerate_vocab);
        if (id >= uni_.size()) uni_.resize(id + 1, Entry{0.f, -0.0f});
        uni_[id] = e;
      } else {
        std::string key;
        for (int i = 0; i < section; ++i) append(key, vocab_.Index(StringPiece(tok[1 + i])));
        higher_.emplace(key, e);
      }
    }
    order_ = (unsigned char)counts.size();
    if (order_ == 0 || order_ > KENLM_MAX_ORDER) die("unsupported model order in", path);
  }

  unsigned char Order() const override { return order_; }
  const base::Vocabulary &BaseVocabulary() const override { return vocab_; }
  void NullContextWrite(void *to) const override { static_cast<State *>(to)->length = 0; }

  float BaseScore(const void *in_state, const WordIndex w, void *out_state) const override {
    const State &in = *static_cast<const State *>(in_state);
    State out;
    const Entry &u = uni_[w < uni_.size() ? w : 0];
    float prob = u.prob;
    out.words[0] = w;
    out.backoff[0] = u.backoff;
    int matched = 1;
    for (int k = 1; k <= (int)in.length && k + 1 <= (int)order_; ++k) {
      std::string key;
      for (int i = k - 1; i >= 0; --i) append(key, in.words[i]);  // chronological order
      append(key, w);
      auto it = higher_.find(key);
      if (k + 1 < (int)order_) out.backoff[k] = -0.0f;
      // An n-gram whose own suffix is not listed (kenlm's test.arpa has "also would consider" without "would consider")
      // is still found: kenlm fills such gaps at load time with the backed-off value, which is what continuing the
      // search past the gap and adding the back-off weights below amounts to.
      if (it == higher_.end()) continue;
      prob = it->second.prob;
      if (k + 1 < (int)order_) out.backoff[k] = it->second.backoff;
      matched = k + 1;
    }
    float ret = prob;
    for (int i = matched - 1; i < (int)in.length; ++i) ret += in.backoff[i];  // float32, shorter context first
    const int keep = matched < (int)order_ - 1 ? matched : (int)order_ - 1;
    for (int i = 1; i < keep; ++i) out.words[i] = in.words[i - 1];
    out.length = (unsigned char)keep;
    *static_cast<State *>(out_state) = out;
    return ret;
  }

 private:
  static void append(std::string &key, WordIndex id) { key.append(reinterpret_cast<const char *>(&id), sizeof(id)); }
  [[noreturn]] static void die(const char *what, const char *path) {
    std::fprintf(stderr, "kenlm stand-in: %s %s\n", what, path);
    std::abort();
  }
  ShimVocabulary vocab_;
  std::vector<Entry> uni_;
  std::unordered_map<std::string, Entry> higher_;
  unsigned char order_ = 0;
};

inline base::Model *LoadVirtual(const char *file_name, const Config &config = Config()) { return new ShimModel(file_name, config); }

}  // namespace ngram
}  // namespace lm
