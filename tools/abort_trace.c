/* Debugging aid (not part of the product or the tests): LD_PRELOAD this to get the native call stack of a process that
 * dies by abort() - pytest's fd capture swallows what the HIP runtime prints just before.  Usage on a GPU box:
 *   gcc -shared -fPIC -o /tmp/abort_trace.so tools/abort_trace.c
 *   XSI_ABORT_TRACE=gpurun_out/abort.txt LD_PRELOAD=/tmp/abort_trace.so python -m pytest tests -x -q -m gpu
 * Also keeps a copy of everything written to fd 2 (the runtime's own message) by pointing fd 2 of the C library's
 * stderr at the same file when XSI_ABORT_STDERR=1. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int g_fd = -1;

static void on_abort(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    if (g_fd >= 0) {
        const char* m = "---- SIGABRT, native stack ----\n";
        (void)!write(g_fd, m, strlen(m));
        backtrace_symbols_fd(frames, n, g_fd);
        fsync(g_fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void init(void) {
    const char* p = getenv("XSI_ABORT_TRACE");
    if (!p) return;
    g_fd = open(p, O_WRONLY | O_CREAT | O_APPEND, 0644);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_abort;
    sigaction(SIGABRT, &sa, NULL);
}

/* the HIP / ROCr runtimes report a fault with fprintf(stderr, ...) right before abort(): interpose fprintf-family
 * writes to stderr is overkill - instead stderr's buffer is flushed to our file by wrapping abort() itself */
void abort(void) {
    if (g_fd >= 0) {
        const char* m = "---- abort() called ----\n";
        (void)!write(g_fd, m, strlen(m));
        void* frames[64];
        int n = backtrace(frames, 64);
        backtrace_symbols_fd(frames, n, g_fd);
        fsync(g_fd);
    }
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
    _exit(134);
}
