/* LD_PRELOAD helper for the GPU box (not a test, not shipped): prints the native call stack of a process that dies by SIGABRT / SIGSEGV -- the box has no
 * debugger.  gcc -shared -fPIC -O1 -o tests/tools/libabrttrace.so tests/tools/abrt_trace.c ; offsets inside libavifhip.so resolve with llvm-symbolizer here. */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_signal(int sig)
{
    void * frames[64];
    const char msg[] = "\n== native stack at the fatal signal ==\n";
    write(2, msg, sizeof(msg) - 1);
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_signal;
    sigaction(SIGABRT, &sa, NULL);
    sigaction(SIGSEGV, &sa, NULL);
}
