// The host layer's fatal-error path (host/ORBextractor.cc `fatal`, host/ORBextractor.h SetFatalErrorHandler): a drop-in
// ORBextractor that cannot get its device calls the application's handler first, then takes the default action.
// usage: fatal_callback <mode>   mode: exit | return | throw | none | abi
//   exit    the handler "saves the map" (prints) and ends the process with its own status 42
//   return  the handler returns: default action = message on cerr + exit(-1)
//   throw   MSORB_THROW=1 is set by the test: the handler returns, std::runtime_error is caught here -> status 7
//   none    no handler registered: message + exit(-1), as in rounds 1-4
//   abi     prints the ABI version of the library and of the header this file was compiled against
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <unistd.h>

#include <opencv2/opencv.hpp>
#include "ORBextractor.h"
#include "msorb.h"

static void handler_exit(int code, const char* what, void* user) {
    std::printf("HANDLER code=%d user=%s what=%s\n", code, (const char*)user, what);
    std::fflush(stdout);
    _exit(42);
}
static void handler_return(int code, const char* what, void* user) {
    std::printf("HANDLER code=%d user=%s what=%s\n", code, (const char*)user, what);
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "none";
    static char tag[] = "atlas";
    if (!std::strcmp(mode, "abi")) {
        std::printf("lib=%d header=%d compatible=%d older_minor=%d next_major=%d\n", msorb_abi_version(), MSORB_ABI_VERSION,
                    msorb_abi_compatible(MSORB_ABI_VERSION), msorb_abi_compatible(MSORB_ABI_VERSION + 1),
                    msorb_abi_compatible(MSORB_ABI_VERSION + 1000));
        return 0;
    }
    if (!std::strcmp(mode, "exit")) ORB_SLAM3::msorb_host::SetFatalErrorHandler(handler_exit, tag);
    if (!std::strcmp(mode, "return") || !std::strcmp(mode, "throw")) ORB_SLAM3::msorb_host::SetFatalErrorHandler(handler_return, tag);
    try {
        ORB_SLAM3::ORBextractor ex(1000, 1.2f, 8, 20, 7);   // MSORB_DEVICE=4096 from the test: no such device
        std::printf("constructed\n");
    } catch (const std::runtime_error& e) {
        std::printf("CAUGHT %s\n", e.what());
        return 7;
    }
    return 0;
}
