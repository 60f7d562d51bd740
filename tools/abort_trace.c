// tools/abort_trace.c — LD_PRELOAD helper for the GPU box: when some thread of the process calls abort() (SIGABRT) or faults (SIGSEGV / SIGBUS), the stack of THAT
// thread goes to the file named by ABORT_TRACE_FILE before the default action runs. Written to find who aborts one run in ten of the GPU tier (profiles/r06/README.md).
//   gcc -shared -fPIC -o tools/ab/abort_trace.so tools/abort_trace.c
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void handler(int sig) {
    const char *path = getenv("ABORT_TRACE_FILE");
    int fd = open(path ? path : "/tmp/abort_trace.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        char head[64];
        int n = snprintf(head, sizeof head, "signal %d, thread stack:\n", sig);
        if (write(fd, head, (size_t)n) < 0) {}
        void *bt[64];
        int k = backtrace(bt, 64);
        backtrace_symbols_fd(bt, k, fd);
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}
__attribute__((constructor)) static void init(void) {
    static char stack[1 << 16];
    stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = handler;
    sa.sa_flags = SA_ONSTACK | SA_NODEFER;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
