// vmtl_conv3x3_small instantiations for 32 input storage channels (one translation unit per channel count so
// that the 12 kernel variants each needs compile in parallel); see conv_small.h.
#include "conv_small.h"

int vmtl_small_launch_cs32(SmallP& p, hipStream_t st) { return launch_small_cs<32>(p, st); }
