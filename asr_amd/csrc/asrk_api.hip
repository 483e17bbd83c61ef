// Version / error-string entry points of libasrk.so.
#include "asrk_common.h"

extern "C" int asrk_version(void) { return ASRK_VERSION; }

extern "C" const char* asrk_strerror(int code) {
    switch (code) {
        case ASRK_OK: return "ok";
        case ASRK_ERR_NULL: return "required pointer is NULL";
        case ASRK_ERR_SHAPE: return "unsupported or inconsistent shape";
        case ASRK_ERR_ALIGN: return "pointer is not 16-byte aligned";
        case ASRK_ERR_HIP: return "HIP runtime call or kernel launch failed";
        case ASRK_ERR_WORKSPACE: return "workspace too small";
        case ASRK_ERR_UNSUPPORTED: return "dtype or mode not implemented";
        case ASRK_ERR_LENGTH: return "mel input features must be 3000 frames long";
        default: return "unknown asrk error";
    }
}
