#include "osd_common.h"
thread_local char g_osd_err[512] = "";
extern "C" const char* osd_last_error_string(void) { return g_osd_err; }
extern "C" int osd_abi_version(void) { return 3; }   // 3: the ordered-mode scratch of the weight-gradient entries travels in osd_conv_desc (no per-stream registry)
