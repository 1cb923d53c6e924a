#include "ia_common.h"
extern "C" const char* ia_version(void) { return "indicasr-hip gfx950 r3"; }
