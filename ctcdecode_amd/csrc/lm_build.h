// lm_build.h -- host side of the LM tier: reads an ARPA language model and the label set and builds the flat tables of
// lm_tables.h.  Counterpart of Scorer::setup (ctcdecode/src/scorer.cpp:43-72: load_lm, :148-161: set_char_map,
// :196-230: fill_dictionary + decoder_utils.cpp:147-193) and of the third-party loader it calls (kenlm's ARPA reader,
// lm/read_arpa.cc: float32 weights, "<unk>" = word 0, the other words numbered in file order, a zero back-off weight
// is immaterial).  Host-only; no HIP types.  Binary kenlm files are not supported (kenlm's source is not available here).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "lm_tables.h"

namespace ctclm {

struct HostScorer {
  double alpha = 0, beta = 0;
  int order = 0;
  bool char_based = true;
  int space_id = -1;
  int dict_size = 0;
  std::vector<std::string> labels;
  std::unordered_map<std::string, uint32_t> word_id;  // LM vocabulary ("<unk>" is 0 and not in the map)
  std::vector<std::string> vocab;                     // in kenlm's enumeration order: <unk>, then file order
  // tables (host copies; `blob` is what goes to HBM, offsets in bytes)
  std::vector<float> uni_prob, st_bo;
  std::vector<uint32_t> uni_state, st_fail, label_word;
  std::vector<NgSlot> ng;
  std::vector<DictNode> dict;
  std::vector<uint32_t> dict_lab;  // wide dictionaries (more than 64 labels): arc labels
  bool dict_wide = false;
  uint32_t s0 = 0, w_bos = 0, w_eos = 0;
  int clean0 = 0;
  std::string error;

  uint32_t id_of(const std::string &w) const {
    auto it = word_id.find(w);
    return it == word_id.end() ? 0u : it->second;
  }

  static std::vector<std::string> utf8_chars(const std::string &s) {  // decoder_utils.cpp:83-100
    std::vector<std::string> r;
    std::string cur;
    for (char c : s) {
      if ((c & 0xc0) != 0x80 && !cur.empty()) {
        r.push_back(cur);
        cur.clear();
      }
      cur.append(1, c);
    }
    r.push_back(cur);
    return r;
  }
  static size_t utf8_len(const std::string &s) {  // decoder_utils.cpp:75-81
    size_t n = 0;
    for (char c : s) n += ((c & 0xc0) != 0x80);
    return n;
  }

  LmView view() const {  // over the host copies
    LmView v;
    v.uni_prob = uni_prob.data(); v.uni_state = uni_state.data(); v.st_bo = st_bo.data(); v.st_fail = st_fail.data();
    v.ng = ng.data(); v.dict = dict.data(); v.label_word = label_word.data();
    v.dict_lab = dict_lab.data(); v.dict_wide = dict_wide ? 1 : 0;
    v.ng_mask = (uint32_t)ng.size() - 1; v.order = order; v.char_based = char_based ? 1 : 0; v.space_id = space_id;
    v.s0 = s0; v.clean0 = clean0; v.w_bos = w_bos; v.w_eos = w_eos; v.alpha = alpha; v.beta = beta;
    v.cb = 0; v.cb_miss = nullptr; v.cb_count = nullptr; v.cb_cap = 0; v.cb_ring = 0;
    return v;
  }

  bool fail(const std::string &msg) {
    error = msg;
    return false;
  }

  bool build(double alpha_, double beta_, const std::string &lm_path, const std::vector<std::string> &labels_) {
    alpha = alpha_; beta = beta_; labels = labels_;
    std::ifstream in(lm_path);
    if (!in) return fail("Invalid language model path: " + lm_path);  // scorer.cpp:57
    {
      char magic[8] = {0};
      in.read(magic, 6);
      if (std::memcmp(magic, "mmap l", 6) == 0) return fail("binary kenlm models are not supported by this build; give the ARPA text file");
      in.clear();
      in.seekg(0);
    }
    // ---- ARPA text
    struct Gram { std::vector<uint32_t> w; float prob, bo; };
    std::vector<std::vector<Gram>> grams(1);
    vocab.push_back("<unk>");
    std::vector<float> up(1, -100.0f), ub(1, 0.0f);  // <unk> when the file lists none (kenlm: unknown_missing_logprob)
    std::string line;
    int section = 0, norders = 0;
    bool data = false;
    while (std::getline(in, line)) {
      while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
      if (line.empty()) continue;
      if (line == "\\data\\") { data = true; continue; }
      if (!data) continue;
      if (line.compare(0, 6, "ngram ") == 0) { ++norders; continue; }
      if (line == "\\end\\") break;
      if (line[0] == '\\') {
        section = std::atoi(line.c_str() + 1);
        if (section < 1 || section > norders) return fail("bad section header in " + lm_path);
        if ((int)grams.size() <= section) grams.resize(section + 1);
        continue;
      }
      if (section == 0) return fail("n-gram line before a section header in " + lm_path);
      std::vector<std::string> f;
      {
        std::istringstream ss(line);
        for (std::string t; ss >> t;) f.push_back(t);
      }
      if ((int)f.size() < section + 1 || (int)f.size() > section + 2) return fail("malformed n-gram line in " + lm_path + ": " + line);
      Gram g;
      g.prob = std::strtof(f[0].c_str(), nullptr);
      if (g.prob > 0.0f) return fail("positive log probability in " + lm_path);
      g.bo = (int)f.size() == section + 2 ? std::strtof(f[section + 1].c_str(), nullptr) : 0.0f;
      if (section == 1) {
        uint32_t id = 0;
        if (f[1] != "<unk>") {
          auto it = word_id.find(f[1]);
          if (it == word_id.end()) {
            id = (uint32_t)vocab.size();
            word_id.emplace(f[1], id);
            vocab.push_back(f[1]);
            up.push_back(0.f);
            ub.push_back(0.f);
          } else {
            id = it->second;
          }
        }
        up[id] = g.prob;
        ub[id] = g.bo;
      } else {
        for (int i = 0; i < section; ++i) g.w.push_back(id_of(f[1 + i]));
        grams[section].push_back(std::move(g));
      }
    }
    order = norders;
    if (order < 1 || order > kMaxOrder) return fail("language model order must be 1.." + std::to_string(kMaxOrder));
    grams.resize(order + 1);
    // ---- states: 0 = empty context, 1 + w = unigram w, then the listed n-grams of orders 2 .. N-1
    const uint32_t W = (uint32_t)vocab.size();
    std::map<std::vector<uint32_t>, uint32_t> state_of;
    uni_prob = up;
    uni_state.resize(W);
    st_bo.assign(1 + W, 0.0f);
    st_fail.assign(1 + W, 0u);
    for (uint32_t w = 0; w < W; ++w) {
      uni_state[w] = order == 1 ? 0u : 1 + w;  // (a unigram model keeps no context: kenlm's state has length 0, no back-off is ever added)
      st_bo[1 + w] = ub[w];
      state_of[{w}] = 1 + w;
    }
    for (int n = 2; n <= order - 1; ++n)
      for (const Gram &g : grams[n]) {
        if (state_of.count(g.w)) continue;
        state_of[g.w] = (uint32_t)st_bo.size();
        st_bo.push_back(g.bo);
        st_fail.push_back(0u);
      }
    auto longest_suffix_state = [&](const std::vector<uint32_t> &g, size_t max_len) -> uint32_t {
      for (size_t len = std::min(max_len, g.size()); len >= 1; --len) {
        auto it = state_of.find(std::vector<uint32_t>(g.end() - len, g.end()));
        if (it != state_of.end()) return it->second;
      }
      return 0u;
    };
    for (const auto &kv : state_of)
      if (kv.first.size() >= 2) st_fail[kv.second] = longest_suffix_state(kv.first, kv.first.size() - 1);
    // ---- (context state, word) -> {prob, next state} for every listed n-gram of order >= 2
    size_t nhigher = 0;
    for (int n = 2; n <= order; ++n) nhigher += grams[n].size();
    size_t cap = 16;
    // (at most a quarter of the slots are taken: most queries are misses, which probe until they meet an empty slot -- 1.4
    //  dependent accesses at this load against 2.5 at one half; measured -1 % (word model) / -6 % (character model) per frame)
    // ... while the table stays small (256 MB: 16 M slots of 16 B); a model of production size keeps the usual half-full table
    // instead of 6-13 GB of slots, mirrored on the host while it is built
    const size_t kQuarterFullUpTo = (size_t)16 << 20;
    while (cap < 4 * nhigher + 2 && cap < kQuarterFullUpTo) cap <<= 1;
    while (cap < 2 * nhigher + 2) cap <<= 1;
    ng.assign(cap, NgSlot{kEmptySlot, 0, 0, 0});
    for (int n = 2; n <= order; ++n)
      for (const Gram &g : grams[n]) {
        const std::vector<uint32_t> ctx(g.w.begin(), g.w.end() - 1);
        auto ci = state_of.find(ctx);
        if (ci == state_of.end()) return fail("n-gram whose context is not listed (kenlm rejects such a model): " + lm_path);
        NgSlot s;
        s.state = ci->second;
        s.word = g.w.back();
        std::memcpy(&s.prob_bits, &g.prob, 4);
        s.next = longest_suffix_state(g.w, (size_t)order - 1);  // the n-gram itself when its order is < N
        uint32_t h = ng_hash(s.state, s.word) & (uint32_t)(cap - 1);
        while (ng[h].state != kEmptySlot) {
          if (ng[h].state == s.state && ng[h].word == s.word) break;  // listed twice: the later line wins
          h = (h + 1) & (uint32_t)(cap - 1);
        }
        ng[h] = s;
      }
    // ---- Scorer::load_lm's character-model test and the special words
    for (const std::string &w : vocab)
      if (w != "<unk>" && w != "<s>" && w != "</s>" && utf8_len(w) > 1) char_based = false;
    w_bos = id_of("<s>");
    w_eos = id_of("</s>");
    {  // state / clean counter after the (N-1) "<s>" tokens a short history is padded with (scorer.cpp:184-189)
      const LmView v = view();
      uint32_t st = 0;
      int cl = 0;
      for (int i = 0; i < order - 1; ++i) lm_cond(v, &st, &cl, w_bos);  // (the values of these windows are not used)
      s0 = st;
      clean0 = cl;
    }
    return build_labels_and_dictionary();
  }

  // The label map and, for word models, the dictionary -- from `labels`, `vocab`, `word_id` and `char_based` (shared with
  // the callback scorer, lm_callback.h, whose vocabulary comes from the caller instead of an ARPA file).
  bool build_labels_and_dictionary() {
    // ---- labels (set_char_map, scorer.cpp:148-161): the last label that reads " " is the space
    std::unordered_map<std::string, int> cmap;
    for (size_t i = 0; i < labels.size(); ++i) {
      if (labels[i] == " ") space_id = (int)i;
      cmap[labels[i]] = (int)i;
    }
    label_word.assign(labels.size(), 0u);
    for (size_t i = 0; i < labels.size(); ++i) label_word[i] = id_of(labels[i]);
    // ---- dictionary of a word model (fill_dictionary(true), scorer.cpp:196-230)
    dict.clear();
    dict_size = 0;
    if (!char_based) {
      if (space_id < 0) return fail("a word language model needs a \" \" label");
      dict_wide = labels.size() > 64;  // more labels than a node's 64-bit arc mask has bits: sorted arc lists instead
      dict_lab.clear();
      struct TNode { std::map<int, int> kids; uint32_t word = kNoWord; };
      std::vector<TNode> trie(1);
      for (const std::string &w : vocab) {  // add_word_to_dictionary, decoder_utils.cpp:164-193
        std::vector<int> ids;
        bool ok = true;
        for (const std::string &ch : utf8_chars(w)) {
          if (ch == " ") { ids.push_back(space_id); continue; }
          auto it = cmap.find(ch);
          if (it == cmap.end()) { ok = false; break; }
          ids.push_back(it->second);
        }
        if (!ok) continue;
        int st = 0;
        for (int id : ids) {
          auto it = trie[st].kids.find(id);
          if (it == trie[st].kids.end()) {
            trie.emplace_back();
            it = trie[st].kids.emplace(id, (int)trie.size() - 1).first;
          }
          st = it->second;
        }
        trie[st].word = id_of(w);  // the word whose spelling ends here ("<unk>" spelled out is word 0 = unknown)
        ++dict_size;
      }
      // breadth-first renumbering: the children of a node are contiguous, in label order; the trailing " " of a word is
      // an arc too (it has a slot that no prefix ever occupies: a completed word restarts at the root)
      std::vector<int> order_bfs(1, 0), new_id(trie.size(), -1);
      new_id[0] = 0;
      dict.assign(1, DictNode{0, 0, 0, kNoWord});
      for (size_t qi = 0; qi < order_bfs.size(); ++qi) {
        const int t = order_bfs[qi];
        DictNode dn{dict_wide ? (uint32_t)dict_lab.size() : 0u, 0, (uint32_t)dict.size(), trie[t].word};
        std::vector<int> labs;
        for (const auto &kv : trie[t].kids) labs.push_back(kv.first);
        if (trie[t].word != kNoWord && !trie[t].kids.count(space_id)) labs.push_back(space_id);
        std::sort(labs.begin(), labs.end());
        for (int lab : labs) {
          if (dict_wide) { dict_lab.push_back((uint32_t)lab); ++dn.mask_hi; }
          else if (lab < 32) dn.mask_lo |= 1u << lab; else dn.mask_hi |= 1u << (lab - 32);
          dict.push_back(DictNode{0, 0, 0, kNoWord});
          auto it = trie[t].kids.find(lab);
          if (it != trie[t].kids.end()) {
            new_id[it->second] = (int)dict.size() - 1;
            order_bfs.push_back(it->second);
          }
        }
        dict[new_id[t]] = dn;
      }
    } else {
      dict.assign(1, DictNode{0, 0, 0, kNoWord});
    }
    return true;
  }

  // ... the same query in the form the host-side scorer hook speaks (lm_callback.h): the float32 log10 probability kenlm's
  // BaseScore returns for the last word, before the reference divides it by NUM_FLT_LOGE; returns 1 for a window with an
  // unknown word (the reference's OOV_SCORE), else 0
  int cond_log10(const std::vector<std::string> &words, float *p10) const {
    const LmView v = view();
    uint32_t st = 0;
    float p = 0.f;
    for (const std::string &w : words) {
      const uint32_t id = id_of(w);
      if (id == 0) return 1;
      p = lm_score(v, st, id, &st);
    }
    *p10 = p;
    return 0;
  }
  // Scorer::get_log_cond_prob (scorer.cpp:74-93) on explicit words -- test / introspection entry
  double cond_log_prob(const std::vector<std::string> &words) const {
    const LmView v = view();
    uint32_t st = 0;
    int cl = order - 1;  // the call starts from the empty context: nothing unknown has been seen
    double r = 0.0;
    for (const std::string &w : words) {
      const uint32_t id = id_of(w);
      if (id == 0) return kOovScore;
      r = lm_cond(v, &st, &cl, id);
    }
    return r;
  }
};

}  // namespace ctclm
