/* par.c -- parallel-for on a persistent pthread pool (host-side stages only).
 *
 * mga_parallel_for(T, n, f, data) runs f(data, i, tid) for i in [0,n) on up to T threads, tid in [0,T) being
 * unique among the threads working on THIS call.  The pool grows on demand and its workers sleep on a condition
 * variable between calls; several pipeline threads may issue calls concurrently.  A call posts T-1 tickets and
 * the caller works as tid 0, so a call never waits for a pool worker to become free before making progress. */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include "mga_host.h"

typedef struct pf_job_s {
	mga_for_f f; void *data; int64_t n, chunk;
	volatile int64_t next;
	volatile int pending;          /* tickets not yet finished */
	pthread_mutex_t mtx; pthread_cond_t done;
} pf_job_t;

typedef struct { pf_job_t *job; int tid; } pf_ticket_t;

#define PF_QCAP 4096
static pthread_mutex_t g_qmtx = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_qcond = PTHREAD_COND_INITIALIZER;
static pf_ticket_t g_q[PF_QCAP];
static int g_qhead = 0, g_qtail = 0, g_n_workers = 0, g_idle = 0;

static void pf_run(pf_job_t *p, int tid)
{
	for (;;) {
		int64_t b = __sync_fetch_and_add(&p->next, p->chunk), e, i;
		if (b >= p->n) break;
		e = b + p->chunk < p->n ? b + p->chunk : p->n;
		for (i = b; i < e; ++i) p->f(p->data, i, tid);
	}
}

static void *pf_worker(void *a)
{
	(void)a;
	{ /* MGA_POOL_NICE=n: the pool's workers run at nice n -- bulk host work (GAF text, chain records) then yields to the pipeline threads, whose kernel launches and stream waits
	   * are what keeps the GPU fed when a rank has one or two cores to itself */
		const char *e = getenv("MGA_POOL_NICE");
		if (e && atoi(e) > 0) (void)setpriority(PRIO_PROCESS, (id_t)syscall(SYS_gettid), atoi(e));
	}
	pthread_mutex_lock(&g_qmtx);
	for (;;) {
		pf_ticket_t t;
		while (g_qhead == g_qtail) { ++g_idle; pthread_cond_wait(&g_qcond, &g_qmtx); --g_idle; }
		t = g_q[g_qhead]; g_qhead = (g_qhead + 1) % PF_QCAP;
		pthread_mutex_unlock(&g_qmtx);
		pf_run(t.job, t.tid);
		pthread_mutex_lock(&t.job->mtx);
		if (--t.job->pending == 0) pthread_cond_signal(&t.job->done);
		pthread_mutex_unlock(&t.job->mtx);
		pthread_mutex_lock(&g_qmtx);
	}
	return 0;
}

void mga_parallel_for(int n_threads, int64_t n, mga_for_f f, void *data)
{
	pf_job_t p;
	int t, n_tick;
	if (n <= 0) return;
	if (n_threads < 1) n_threads = 1;
	if ((int64_t)n_threads > n) n_threads = (int)n;
	p.f = f, p.data = data, p.n = n, p.next = 0;
	p.chunk = n / ((int64_t)n_threads * 16) > 0 ? n / ((int64_t)n_threads * 16) : 1;
	if (p.chunk > 64) p.chunk = 64;
	if (n_threads == 1) { pf_run(&p, 0); return; }
	n_tick = n_threads - 1;
	p.pending = n_tick;
	pthread_mutex_init(&p.mtx, 0); pthread_cond_init(&p.done, 0);
	pthread_mutex_lock(&g_qmtx);
	{ /* enough workers for the tickets of all concurrent calls */
		int queued = (g_qtail - g_qhead + PF_QCAP) % PF_QCAP, want = queued + n_tick - g_idle;
		while (want > 0 && g_n_workers < 1024) {
			pthread_t th;
			pthread_attr_t at;
			pthread_attr_init(&at); pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
			if (pthread_create(&th, &at, pf_worker, 0) != 0) { pthread_attr_destroy(&at); break; }
			pthread_attr_destroy(&at);
			++g_n_workers, --want;
		}
	}
	for (t = 0; t < n_tick; ++t) {
		if ((g_qtail + 1) % PF_QCAP == g_qhead) { --p.pending; continue; } /* queue full: the caller and the posted tickets do the work */
		g_q[g_qtail].job = &p, g_q[g_qtail].tid = t + 1; g_qtail = (g_qtail + 1) % PF_QCAP;
	}
	pthread_cond_broadcast(&g_qcond);
	pthread_mutex_unlock(&g_qmtx);
	pf_run(&p, 0);
	pthread_mutex_lock(&p.mtx);
	while (p.pending > 0) pthread_cond_wait(&p.done, &p.mtx);
	pthread_mutex_unlock(&p.mtx);
	pthread_mutex_destroy(&p.mtx); pthread_cond_destroy(&p.done);
}
