/* par.c -- minimal parallel-for on pthreads with an atomic work counter (host-side stages only). */
#include <pthread.h>
#include "mga_host.h"

typedef struct { mga_for_f f; void *data; int64_t n; volatile int64_t next; int64_t chunk; } pf_t;
typedef struct { pf_t *p; int tid; } pf_w_t;

static void *pf_worker(void *a)
{
	pf_w_t *w = (pf_w_t*)a;
	pf_t *p = w->p;
	for (;;) {
		int64_t b = __sync_fetch_and_add(&p->next, p->chunk), e, i;
		if (b >= p->n) break;
		e = b + p->chunk < p->n ? b + p->chunk : p->n;
		for (i = b; i < e; ++i) p->f(p->data, i, w->tid);
	}
	return 0;
}

void mga_parallel_for(int n_threads, int64_t n, mga_for_f f, void *data)
{
	pf_t p;
	int t;
	if (n <= 0) return;
	if (n_threads < 1) n_threads = 1;
	p.f = f, p.data = data, p.n = n, p.next = 0;
	p.chunk = n / (n_threads * 16) > 0 ? n / (n_threads * 16) : 1;
	if (p.chunk > 64) p.chunk = 64;
	if (n_threads == 1) { pf_w_t w = { &p, 0 }; pf_worker(&w); return; }
	{
		pthread_t *tid = MGA_MALLOC(pthread_t, n_threads);
		pf_w_t *w = MGA_MALLOC(pf_w_t, n_threads);
		for (t = 0; t < n_threads; ++t) { w[t].p = &p, w[t].tid = t; pthread_create(&tid[t], 0, pf_worker, &w[t]); }
		for (t = 0; t < n_threads; ++t) pthread_join(tid[t], 0);
		free(tid); free(w);
	}
}
