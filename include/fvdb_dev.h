/* fvdb_dev.h — development-only entry points.  NOT part of the product ABI: libfvdb_hip.so does not export them; they exist in
 * the dev build only (`make -C fabstir-vectordb_amd dev` -> lib_dev/libfvdb_hip.so, compiled with -DFVDB_DEV_TOOLS), which
 * tools/emulate_rank.py loads through FVDB_LIB_DIR. */
#ifndef FVDB_DEV_H
#define FVDB_DEV_H
#include "fvdb.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Capacity planning on ONE GPU: a communicator of `world` ranks whose exchanges are device-to-device copies of this
 * rank's own blocks (every peer is pretended to have sent what this rank sent).  Results are meaningless; the step's
 * kernels, buffer sizes and stream ordering are exactly those of rank `rank` in a real `world`-rank job, so its time is
 * the per-rank step time less the fabric.  bench.py --emulate-world N. */
int fvdb_comm_create_loopback(fvdb_ctx* ctx, int world, int rank, fvdb_comm** out);
#ifdef __cplusplus
}
#endif
#endif /* FVDB_DEV_H */
