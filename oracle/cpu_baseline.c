/*
 * cpu_baseline.c -- the reference's CPU search structure, for bench.py's cpu_baseline leg only
 * (TEST INFRASTRUCTURE; kind = "port": the C# reference cannot run on this box).
 *
 * Structure follows BaseSLAM/ParallelWorker.cs:34-117 and CoreSLAM/CoreSLAMProcessor.cs:674-710:
 *   - T persistent threads created once (ParallelWorker ctor :34-56), each blocked on its own
 *     signal (SignalConcurrentQueue.EnqueuedItemSignal, :70-74);
 *   - Work(action, wait=true) hands the same action to every thread and blocks until all have
 *     signalled completion (:98-117);
 *   - every thread runs MonteCarloSearch over ITS OWN pre-drawn offsets (the per-thread random
 *     queues, CoreSLAMProcessor.cs:680-689), base pose evaluated once per thread (:627);
 *   - the caller does the serial arg-min over the T results with strict '<' (:695-705).
 * The per-candidate arithmetic is oracle_cs_distance (scalar, like the live SISD path).
 */
#include "oracle.h"
#include <pthread.h>
#include <stdlib.h>
#include <time.h>
#include <limits.h>

typedef struct {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv_start, cv_done;
    int has_work, done, quit, index;
    struct job *job;
} worker_t;

struct job {
    const uint16_t *pixels; int size; float scale;
    const float *xy; int n_points;
    const float *search_pose; const float *offs; int iters;
    int32_t *dist; int32_t *best_local;   /* per thread */
};

static void *work_loop(void *arg)      /* ParallelWorker.cs:67-91 */
{
    worker_t *w = (worker_t *)arg;
    for (;;) {
        pthread_mutex_lock(&w->mu);
        while (!w->has_work && !w->quit) pthread_cond_wait(&w->cv_start, &w->mu);
        if (w->quit) { pthread_mutex_unlock(&w->mu); return NULL; }
        w->has_work = 0;
        struct job *j = w->job;
        pthread_mutex_unlock(&w->mu);

        /* CoreSLAMProcessor.cs:682-688: MonteCarloSearch with this thread's queue */
        int32_t d, bi;
        bi = oracle_cs_search(j->pixels, j->size, j->scale, j->xy, j->n_points, j->search_pose,
                              j->offs + 3 * (size_t)w->index * j->iters, j->iters, NULL, &d, NULL);
        j->dist[w->index] = d;
        j->best_local[w->index] = bi;

        pthread_mutex_lock(&w->mu);    /* item.WaitHandle.Set() :85 */
        w->done = 1;
        pthread_cond_signal(&w->cv_done);
        pthread_mutex_unlock(&w->mu);
    }
}

double oracle_cpu_baseline_search(const uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                  const float search_pose[3], const float *offs,
                                  int n_threads, int iters, int n_scans,
                                  int64_t *out_evals, int32_t *out_best_index, int32_t *out_best_dist)
{
    return oracle_cpu_baseline_search_timed(pixels, size, scale, xy, n_points, search_pose, offs, n_threads, iters, n_scans,
                                            out_evals, out_best_index, out_best_dist, NULL);
}

/* the same, with the wall time of every scan in scan_secs[n_scans] (median / p95 of the baseline) */
double oracle_cpu_baseline_search_timed(const uint16_t *pixels, int size, float scale, const float *xy, int n_points,
                                        const float search_pose[3], const float *offs,
                                        int n_threads, int iters, int n_scans,
                                        int64_t *out_evals, int32_t *out_best_index, int32_t *out_best_dist, double *scan_secs)
{
    worker_t *ws = (worker_t *)calloc((size_t)n_threads, sizeof(worker_t));
    int32_t *dist = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_threads);
    int32_t *bl = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_threads);
    struct job jb = { pixels, size, scale, xy, n_points, search_pose, offs, iters, dist, bl };
    for (int i = 0; i < n_threads; i++) {
        ws[i].index = i; ws[i].job = &jb;
        pthread_mutex_init(&ws[i].mu, NULL);
        pthread_cond_init(&ws[i].cv_start, NULL);
        pthread_cond_init(&ws[i].cv_done, NULL);
        pthread_create(&ws[i].th, NULL, work_loop, &ws[i]);
    }
    int32_t best_d = INT32_MAX, best_i = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int sc = 0; sc < n_scans; sc++) {
        struct timespec s0, s1;
        if (scan_secs) clock_gettime(CLOCK_MONOTONIC, &s0);
        for (int i = 0; i < n_threads; i++) {                 /* Work(): enqueue + signal :102-111 */
            pthread_mutex_lock(&ws[i].mu);
            ws[i].has_work = 1; ws[i].done = 0;
            pthread_cond_signal(&ws[i].cv_start);
            pthread_mutex_unlock(&ws[i].mu);
        }
        for (int i = 0; i < n_threads; i++) {                 /* WaitHandle.WaitAll :115 */
            pthread_mutex_lock(&ws[i].mu);
            while (!ws[i].done) pthread_cond_wait(&ws[i].cv_done, &ws[i].mu);
            pthread_mutex_unlock(&ws[i].mu);
        }
        best_d = INT32_MAX; best_i = 0;                       /* CoreSLAMProcessor.cs:695-705 */
        for (int i = 0; i < n_threads; i++)
            if (dist[i] < best_d) {
                best_d = dist[i];
                /* flat index: 0 = base pose, 1 + thread*iters + (local-1) otherwise */
                best_i = bl[i] == 0 ? 0 : 1 + i * iters + (bl[i] - 1);
            }
        if (scan_secs) {
            clock_gettime(CLOCK_MONOTONIC, &s1);
            scan_secs[sc] = (double)(s1.tv_sec - s0.tv_sec) + 1e-9 * (double)(s1.tv_nsec - s0.tv_nsec);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int i = 0; i < n_threads; i++) {
        pthread_mutex_lock(&ws[i].mu);
        ws[i].quit = 1;
        pthread_cond_signal(&ws[i].cv_start);
        pthread_mutex_unlock(&ws[i].mu);
        pthread_join(ws[i].th, NULL);
    }
    if (out_evals) *out_evals = (int64_t)n_scans * n_threads * ((int64_t)iters + 1);
    if (out_best_index) *out_best_index = best_i;
    if (out_best_dist) *out_best_dist = best_d;
    free(ws); free(dist); free(bl);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
