/* LD_PRELOAD helper for chasing a silent abort(): prints the C backtrace of the caller before handing over to libc's abort().
 *   gcc -shared -fPIC -O1 -o /tmp/abort_bt.so tools/dbg/abort_bt.c -ldl
 *   LD_PRELOAD=/tmp/abort_bt.so python -m pytest ...
 * Debug aid only -- nothing in the package or the tests loads it. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int out_fd = 2;      /* C2M_ABORT_BT_FILE: pytest captures fd 2 while a test runs -- write somewhere it does not reach */

static void dump(const char* why) {
    void* frames[64];
    int n = backtrace(frames, 64);
    dprintf(out_fd, "\n==== abort_bt: %s; %d frames ====\n", why, n);
    backtrace_symbols_fd(frames, n, out_fd);
    dprintf(out_fd, "==== end ====\n");
}

void abort(void) {
    dump("abort() called");
    void (*real)(void) = (void (*)(void))dlsym(RTLD_NEXT, "abort");
    if (real) real();
    _exit(134);
}

static void on_sig(int sig) {
    dump(sig == SIGABRT ? "SIGABRT" : "signal");
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void init(void) {
    const char* path = getenv("C2M_ABORT_BT_FILE");
    if (path) {
        int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) out_fd = fd;
    }
    signal(SIGABRT, on_sig);
}
