// Stand-in for kenlm's util/tokenize_piece.hh: included by scorer.cpp:10, nothing from it is used.
#pragma once
