// pending.hip — entry points of include/megagta_hip.h whose kernels are not built yet: they fail
// loudly with MGTA_EUNSUPPORTED (never a silent fallback).  Each one moves to its own .hip when built.
#include "common.hpp"
using namespace mgta;
extern "C" {
int mgta_hmm_load(mgta_ctx *, int, int, const double *, const double *, const double *, const double *, const int32_t *, mgta_hmm **) {
    set_error("mgta_hmm_load: not built yet"); return MGTA_EUNSUPPORTED; }
void mgta_hmm_free(mgta_hmm *) {}
int mgta_astar_batch(mgta_sdbg *, const mgta_hmm *, const mgta_hmm *, const char *, const int32_t *, int64_t, int, double, int,
                     mgta_contig_sink, void *, mgta_astar_stats *) {
    set_error("mgta_astar_batch: not built yet"); return MGTA_EUNSUPPORTED; }
}
