// LD_PRELOAD diagnostic: print a native backtrace when std::terminate runs (used by tools/spawn_teardown_probe.py to find WHICH thread of a
// spawned DataLoader worker ended in "terminate called without an active exception").  g++ -shared -fPIC -O1 -o terminate_trace.so terminate_trace.cpp
#include <exception>
#include <execinfo.h>
#include <unistd.h>
#include <cstdlib>
#include <cstdio>
#include <sys/syscall.h>

static void on_terminate() {
    char head[128];
    int n = snprintf(head, sizeof head, "[terminate_trace] pid %d tid %ld std::terminate, native backtrace:\n", (int)getpid(), (long)syscall(SYS_gettid));
    if (write(2, head, n) < 0) {}
    void* frames[96];
    int depth = backtrace(frames, 96);
    backtrace_symbols_fd(frames, depth, 2);
    abort();
}

__attribute__((constructor)) static void install() { std::set_terminate(on_terminate); }
