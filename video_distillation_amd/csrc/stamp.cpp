// Stamp of the sources libvd_hip.so was built from: hip.py:build compiles this file on every link with
// -DVD_SOURCES_HASH="<sha256 of the kernel sources + include/vd_hip.h, 16 hex digits>"; hip.lib() compares it with the
// checkout's and refuses (or rebuilds) a library built from other sources -- an mtime says nothing after a copy.
#include "../../include/vd_hip.h"
#ifndef VD_SOURCES_HASH
#define VD_SOURCES_HASH "unstamped"
#endif
extern "C" const char* vd_sources_hash(void) { return VD_SOURCES_HASH; }
