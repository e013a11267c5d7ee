/* process-wide knobs of the reference API (reference misc.c:5-7) */
#include "../../include/minigraph_amd.h"
int mg_verbose = 1;
int mg_dbg_flag = 0;
double mg_realtime0 = 0.0;
