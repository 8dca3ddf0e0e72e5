/* Where does OpenJPEG spend an MSB-plane encode?  A 1 kHz SIGPROF sampler around lbdrn_jp2_encode (one thread), program
 * counters written out raw; scripts/jp2_profile.py resolves them against libopenjp2's symbol table and sums them by stage
 * (DWT / tier-1 (context modelling + MQ coder) / tier-2 / rest).  No GPU.  Built and run by scripts/jp2_profile.py:
 *     gcc -O2 -o /tmp/jp2_profile scripts/jp2_profile.c -ldl        ./jp2_profile lib.so planes.raw C H W bits out.txt */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <ucontext.h>

#define MAXS (1 << 20)
static uintptr_t pcs[MAXS];
static volatile int npc = 0;

static void on_prof(int sig, siginfo_t* si, void* ctx)
{
    (void)sig; (void)si;
    ucontext_t* uc = (ucontext_t*)ctx;
    if (npc < MAXS) pcs[npc++] = (uintptr_t)uc->uc_mcontext.gregs[REG_RIP];
}

typedef int (*enc_fn)(const void*, int32_t, int32_t, int32_t, int32_t, uint8_t**, size_t*);
typedef int (*thr_fn)(int32_t);
typedef void (*free_fn)(uint8_t*);

int main(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s liblbdrn_jp2.so planes.raw C H W bits out.txt\n", argv[0]); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    enc_fn enc = (enc_fn)dlsym(h, "lbdrn_jp2_encode");
    thr_fn thr = (thr_fn)dlsym(h, "lbdrn_jp2_set_threads");
    free_fn fr = (free_fn)dlsym(h, "lbdrn_jp2_free");
    const int C = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]), bits = atoi(argv[6]);
    const size_t n = (size_t)C * H * W;
    uint16_t* x = malloc(n * 2);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(x, 2, n, f) != n) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    thr(0);   /* the caller's thread only: every sample is this thread's */
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_prof;
    sa.sa_flags = SA_SIGINFO | SA_RESTART;
    sigaction(SIGPROF, &sa, NULL);
    struct itimerval tv = {{0, 1000}, {0, 1000}};
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    setitimer(ITIMER_PROF, &tv, NULL);
    uint8_t* out = NULL; size_t nb = 0;
    int rc = enc(x, C, H, W, bits, &out, &nb);
    struct itimerval off = {{0, 0}, {0, 0}};
    setitimer(ITIMER_PROF, &off, NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc) { fprintf(stderr, "encode failed\n"); return 1; }
    FILE* o = fopen(argv[7], "w");
    fprintf(o, "seconds %.3f bytes %zu samples %d\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), nb, npc);
    FILE* m = fopen("/proc/self/maps", "r");
    char line[512];
    while (fgets(line, sizeof line, m))
        if (strstr(line, "r-xp") || strstr(line, "r-x")) fprintf(o, "map %s", line);
    fclose(m);
    for (int i = 0; i < npc; ++i) fprintf(o, "pc %lx\n", (unsigned long)pcs[i]);
    fclose(o);
    fr(out);
    return 0;
}
