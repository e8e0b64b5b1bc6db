// Library identity; also the place where the public header is compiled against the
// definitions so a signature drift fails the build instead of the loader.
#include "common.h"
#include "../../include/vmtl.h"

extern "C" const char* vmtl_version(void) { return "vmtl 0.1 (gfx950)"; }
