// Stand-in for kenlm's lm/word_index.hh (kenlm submodule is empty in the reference tree).
#pragma once
namespace lm { typedef unsigned int WordIndex; }
