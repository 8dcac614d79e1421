#include "osd_common.h"
thread_local char g_osd_err[512] = "";
extern "C" const char* osd_last_error_string(void) { return g_osd_err; }
// 3: the ordered-mode scratch of the weight-gradient entries travels in osd_conv_desc (no per-stream registry)
// 4 (round 6): conv algo ids renumbered since 3 (tile 5 = the 128 x 256 LDS-DMA tile, tile 6 / variant 0 = conv_sp's 128 x 128 tile,
//    41 retired), one-pass GroupNorm entries (timeouts poison their outputs, residency checked at launch), the entries rounds 5 - 6 added
extern "C" int osd_abi_version(void) { return 4; }
