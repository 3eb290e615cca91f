/* LD_PRELOAD helper for the GPU box (not a test, not shipped): keeps the last HIP runtime calls of the process -- allocations, frees, copies with
 * their host and device addresses, launches, synchronisations, each with its thread -- in a ring, and the live device / pinned allocations in a
 * table; when the process dies by SIGABRT / SIGSEGV (the HSA runtime aborts after "Memory access fault by GPU ... on address X") it writes both,
 * and /proc/self/maps, to $HIPTRACE_OUT.<pid> (default gpurun_out/hiptrace).  The address of the fault is then looked up in that file: which
 * buffer, allocated by whom, last touched by which copy on which stream.
 *   gcc -shared -fPIC -O1 -o tests/tools/libhiptrace.so tests/tools/hip_trace.c -ldl -lpthread */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

typedef int hipError_t;
typedef void * hipStream_t;
typedef struct { uint32_t x, y, z; } dim3;

enum { RING = 1 << 13, LIVE = 4096, HISTORY = 1 << 14 };
struct Event
{
    double t;
    int tid;
    const char * op;
    const void * a, * b;
    size_t n, m;
    const void * stream;
    int result;
};
static struct Event ring[RING];
static volatile uint64_t ringNext;
struct Live
{
    const void * ptr;
    size_t bytes;
    int tid, pinned;
};
static struct Live live[LIVE];
static pthread_mutex_t liveMutex = PTHREAD_MUTEX_INITIALIZER;

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void note(const char * op, const void * a, const void * b, size_t n, size_t m, const void * stream, int result)
{
    const uint64_t k = __atomic_fetch_add(&ringNext, 1, __ATOMIC_RELAXED);
    struct Event * e = &ring[k % RING];
    e->t = now(), e->tid = (int)syscall(SYS_gettid), e->op = op, e->a = a, e->b = b, e->n = n, e->m = m, e->stream = stream, e->result = result;
}

/* every allocation and release of the process, in order (allocations are few): was the faulting address EVER device memory, and who freed it when */
struct Past
{
    double t;
    const void * ptr;
    size_t bytes;
    int tid, kind; /* 0 device, 1 pinned; +2: release */
};
static struct Past history[HISTORY];
static volatile uint64_t historyNext;
static void remember(const void * p, size_t bytes, int kind)
{
    const uint64_t k = __atomic_fetch_add(&historyNext, 1, __ATOMIC_RELAXED);
    if (k < HISTORY) {
        struct Past * h = &history[k];
        h->t = now(), h->ptr = p, h->bytes = bytes, h->tid = (int)syscall(SYS_gettid), h->kind = kind;
    }
}

static void liveAdd(const void * p, size_t bytes, int pinned)
{
    remember(p, bytes, pinned);
    pthread_mutex_lock(&liveMutex);
    for (int k = 0; k < LIVE; ++k)
        if (!live[k].ptr) {
            live[k].ptr = p, live[k].bytes = bytes, live[k].tid = (int)syscall(SYS_gettid), live[k].pinned = pinned;
            break;
        }
    pthread_mutex_unlock(&liveMutex);
}
static void liveRemove(const void * p)
{
    remember(p, 0, 2);
    pthread_mutex_lock(&liveMutex);
    for (int k = 0; k < LIVE; ++k)
        if (live[k].ptr == p) {
            live[k].ptr = NULL;
            break;
        }
    pthread_mutex_unlock(&liveMutex);
}

/* the runtime is loaded by dlopen (RTLD_LOCAL) when python loads libavifhip.so: not in the scope RTLD_NEXT searches */
static void * resolve(const char * name)
{
    void * p = dlsym(RTLD_NEXT, name);
    if (!p) {
        static void * runtime;
        if (!runtime)
            runtime = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
        if (!runtime)
            runtime = dlopen("libamdhip64.so", RTLD_NOW);
        if (runtime)
            p = dlsym(runtime, name);
    }
    if (!p) {
        static const char msg[] = "hip_trace: cannot resolve a HIP runtime symbol\n";
        (void)!write(2, msg, sizeof(msg) - 1);
        _exit(97);
    }
    return p;
}
#define NEXT(name) \
    static __typeof__(&name) next; \
    if (!next) \
        next = (__typeof__(&name))resolve(#name);

hipError_t hipMalloc(void ** p, size_t bytes)
{
    NEXT(hipMalloc);
    const hipError_t r = next(p, bytes);
    note("hipMalloc", p ? *p : NULL, NULL, bytes, 0, NULL, r);
    if (r == 0 && p)
        liveAdd(*p, bytes, 0);
    return r;
}
hipError_t hipFree(void * p)
{
    NEXT(hipFree);
    note("hipFree>", p, NULL, 0, 0, NULL, 0);
    const hipError_t r = next(p);
    note("hipFree<", p, NULL, 0, 0, NULL, r);
    liveRemove(p);
    return r;
}
hipError_t hipHostMalloc(void ** p, size_t bytes, unsigned flags)
{
    NEXT(hipHostMalloc);
    const hipError_t r = next(p, bytes, flags);
    note("hipHostMalloc", p ? *p : NULL, NULL, bytes, flags, NULL, r);
    if (r == 0 && p)
        liveAdd(*p, bytes, 1);
    return r;
}
hipError_t hipHostFree(void * p)
{
    NEXT(hipHostFree);
    const hipError_t r = next(p);
    note("hipHostFree", p, NULL, 0, 0, NULL, r);
    liveRemove(p);
    return r;
}
hipError_t hipMemcpy(void * d, const void * s, size_t n, int kind)
{
    NEXT(hipMemcpy);
    note("hipMemcpy>", d, s, n, (size_t)kind, NULL, 0);
    const hipError_t r = next(d, s, n, kind);
    note("hipMemcpy<", d, s, n, (size_t)kind, NULL, r);
    return r;
}
hipError_t hipMemcpyAsync(void * d, const void * s, size_t n, int kind, hipStream_t st)
{
    NEXT(hipMemcpyAsync);
    note("hipMemcpyAsync>", d, s, n, (size_t)kind, st, 0);
    const hipError_t r = next(d, s, n, kind, st);
    note("hipMemcpyAsync<", d, s, n, (size_t)kind, st, r);
    return r;
}
hipError_t hipMemcpy2D(void * d, size_t dp, const void * s, size_t sp, size_t w, size_t h, int kind)
{
    NEXT(hipMemcpy2D);
    note("hipMemcpy2D>", d, s, (kind == 1 ? sp : dp) * h, w, NULL, 0);
    const hipError_t r = next(d, dp, s, sp, w, h, kind);
    note("hipMemcpy2D<", d, s, (kind == 1 ? sp : dp) * h, (size_t)kind, NULL, r);
    return r;
}
hipError_t hipMemcpy2DAsync(void * d, size_t dp, const void * s, size_t sp, size_t w, size_t h, int kind, hipStream_t st)
{
    NEXT(hipMemcpy2DAsync);
    /* n: the extent of the HOST side (pitch x rows); m: kind (1 = host to device, 2 = device to host) */
    note("hipMemcpy2DAsync>", d, s, (kind == 1 ? sp : dp) * h, (size_t)kind, st, 0);
    const hipError_t r = next(d, dp, s, sp, w, h, kind, st);
    note("hipMemcpy2DAsync<", d, s, (kind == 1 ? sp : dp) * h, (size_t)kind, st, r);
    return r;
}
hipError_t hipMemsetAsync(void * d, int v, size_t n, hipStream_t st)
{
    NEXT(hipMemsetAsync);
    const hipError_t r = next(d, v, n, st);
    note("hipMemsetAsync", d, NULL, n, 0, st, r);
    return r;
}
hipError_t hipLaunchKernel(const void * f, dim3 g, dim3 b, void ** args, size_t shmem, hipStream_t st)
{
    NEXT(hipLaunchKernel);
    const hipError_t r = next(f, g, b, args, shmem, st);
    note("hipLaunchKernel", f, NULL, (size_t)g.x * g.y * g.z, (size_t)b.x * b.y * b.z, st, r);
    return r;
}
hipError_t hipStreamSynchronize(hipStream_t st)
{
    NEXT(hipStreamSynchronize);
    note("hipStreamSynchronize>", NULL, NULL, 0, 0, st, 0);
    const hipError_t r = next(st);
    note("hipStreamSynchronize<", NULL, NULL, 0, 0, st, r);
    return r;
}
hipError_t hipDeviceSynchronize(void)
{
    NEXT(hipDeviceSynchronize);
    const hipError_t r = next();
    note("hipDeviceSynchronize", NULL, NULL, 0, 0, NULL, r);
    return r;
}
hipError_t hipStreamCreateWithFlags(hipStream_t * st, unsigned flags)
{
    NEXT(hipStreamCreateWithFlags);
    const hipError_t r = next(st, flags);
    note("hipStreamCreateWithFlags", NULL, NULL, flags, 0, st ? *st : NULL, r);
    return r;
}
hipError_t hipStreamDestroy(hipStream_t st)
{
    NEXT(hipStreamDestroy);
    const hipError_t r = next(st);
    note("hipStreamDestroy", NULL, NULL, 0, 0, st, r);
    return r;
}
hipError_t hipHostRegister(void * p, size_t bytes, unsigned flags)
{
    NEXT(hipHostRegister);
    const hipError_t r = next(p, bytes, flags);
    note("hipHostRegister", p, NULL, bytes, flags, NULL, r);
    return r;
}
hipError_t hipHostUnregister(void * p)
{
    NEXT(hipHostUnregister);
    const hipError_t r = next(p);
    note("hipHostUnregister", p, NULL, 0, 0, NULL, r);
    return r;
}

static void out(int fd, const char * fmt, ...)
{
    char line[512];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(line, sizeof(line), fmt, ap);
    va_end(ap);
    if (n > 0)
        (void)!write(fd, line, (size_t)(n < (int)sizeof(line) ? n : (int)sizeof(line) - 1));
}

static void dump(int sig)
{
    char path[256];
    const char * base = getenv("HIPTRACE_OUT");
    snprintf(path, sizeof(path), "%s.%d", base ? base : "gpurun_out/hiptrace", (int)getpid());
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0)
        return;
    out(fd, "signal %d in thread %d at %.6f\n== native stack ==\n", sig, (int)syscall(SYS_gettid), now());
    void * frames[64];
    backtrace_symbols_fd(frames, backtrace(frames, 64), fd);
    out(fd, "== live allocations ==\n");
    for (int k = 0; k < LIVE; ++k)
        if (live[k].ptr)
            out(fd, "%s %p .. %p (%zu bytes) by thread %d\n", live[k].pinned ? "pinned" : "device", live[k].ptr, (const char *)live[k].ptr + live[k].bytes,
                live[k].bytes, live[k].tid);
    out(fd, "== every allocation / release, oldest first ==\n");
    for (uint64_t k = 0; k < historyNext && k < HISTORY; ++k)
        out(fd, "%.6f t%d %s %p .. %p (%zu bytes)\n", history[k].t, history[k].tid, history[k].kind == 0 ? "device" : history[k].kind == 1 ? "pinned" : "release",
            history[k].ptr, (const char *)history[k].ptr + history[k].bytes, history[k].bytes);
    const uint64_t end = ringNext, begin = end > RING ? end - RING : 0;
    out(fd, "== last %llu runtime calls (oldest first) ==\n", (unsigned long long)(end - begin));
    for (uint64_t k = begin; k < end; ++k) {
        const struct Event * e = &ring[k % RING];
        out(fd, "%.6f t%d %s a=%p b=%p n=%zu m=%zu stream=%p -> %d\n", e->t, e->tid, e->op ? e->op : "?", e->a, e->b, e->n, e->m, e->stream, e->result);
    }
    out(fd, "== /proc/self/maps ==\n");
    const int maps = open("/proc/self/maps", O_RDONLY);
    if (maps >= 0) {
        char buf[4096];
        ssize_t n;
        while ((n = read(maps, buf, sizeof(buf))) > 0)
            (void)!write(fd, buf, (size_t)n);
        close(maps);
    }
    close(fd);
}

static void on_signal(int sig)
{
    dump(sig);
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
