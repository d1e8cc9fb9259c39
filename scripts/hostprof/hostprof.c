// Sampling profiler of the host side of a run (diagnostic; there is no perf on the GPU boxes): LD_PRELOAD this library with
// HOSTPROF_OUT=<file>; every HOSTPROF_US microseconds of PROCESS CPU time (default 1000) SIGPROF lands on a running thread and the
// handler records its program counter. At exit the counters go to <file> together with /proc/self/maps; scripts/hostprof/report.py
// turns that into a per-library / per-function table (symbols from the in-tree .so files with nm).
#define _GNU_SOURCE
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <stdint.h>
#include <stdatomic.h>
#include <execinfo.h>
#define DEPTH 16
#define MAXS (1 << 22)
#define MAXST (1 << 18)
static uint64_t* samples;
static uint64_t* stacks;
static atomic_long n_samples;
static void on_prof(int sig, siginfo_t* si, void* uc_) {
    (void)sig; (void)si;
    ucontext_t* uc = (ucontext_t*)uc_;
    long k = atomic_fetch_add(&n_samples, 1);
    if (k < MAXS) samples[k] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
    if (stacks && k < MAXST) {   // HOSTPROF_STACKS=1: call chain through the unwinder (not async-signal-safe in general: diagnostic runs only)
        void* bt[DEPTH + 4];
        int n = backtrace(bt, DEPTH + 4);
        for (int i = 0; i < DEPTH; i++) stacks[k * DEPTH + i] = i + 3 < n ? (uint64_t)bt[i + 3] : 0;
    }
}
__attribute__((constructor)) static void start(void) {
    if (!getenv("HOSTPROF_OUT")) return;
    samples = (uint64_t*)calloc(MAXS, sizeof(uint64_t));
    if (getenv("HOSTPROF_STACKS")) { void* w[4]; backtrace(w, 4); stacks = (uint64_t*)calloc((size_t)MAXST * DEPTH, sizeof(uint64_t)); }
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_prof; sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    long us = getenv("HOSTPROF_US") ? atol(getenv("HOSTPROF_US")) : 1000;
    struct itimerval it; it.it_interval.tv_sec = 0; it.it_interval.tv_usec = us; it.it_value = it.it_interval;
    setitimer(ITIMER_PROF, &it, NULL);
}
__attribute__((destructor)) static void stop(void) {
    const char* out = getenv("HOSTPROF_OUT");
    if (!out || !samples) return;
    struct itimerval it; memset(&it, 0, sizeof(it)); setitimer(ITIMER_PROF, &it, NULL);
    long n = atomic_load(&n_samples); if (n > MAXS) n = MAXS;
    if (n < 100) return;   // wrapper processes (timeout, sh) share the environment: only a process that did work writes
    FILE* f = fopen(out, "w");
    if (!f) return;
    fprintf(f, "samples %ld\n", n);
    FILE* m = fopen("/proc/self/maps", "r");
    char line[1024];
    if (m) { while (fgets(line, sizeof(line), m)) if (strstr(line, " r-xp ") || strstr(line, " r-xs ")) fprintf(f, "map %s", line); fclose(m); }
    for (long i = 0; i < n; i++) {
        fprintf(f, "%lx", (unsigned long)samples[i]);
        if (stacks && i < MAXST) for (int d = 0; d < DEPTH && stacks[i * DEPTH + d]; d++) fprintf(f, " %lx", (unsigned long)stacks[i * DEPTH + d]);
        fprintf(f, "\n");
    }
    fclose(f);
}
