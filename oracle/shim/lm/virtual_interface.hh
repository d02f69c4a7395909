// Stand-in for kenlm's lm/virtual_interface.hh: nothing from it is used on the no-LM path.
#pragma once
