/* lghip_p2p.h - C ABI of the peer-window gradient exchange of liblghip.so: the data-parallel exchange step of
 * BASELINE config #4 written by hand for xGMI, with no collective library in the path.
 *
 * The reference has no distributed code at all; the contract is SURVEY.md 8e: one process per GPU,
 * one exchange of the flat gradient bucket per training step, no host synchronisation, identical update on every rank.
 *
 * How it works.  Every rank owns a WINDOW of uncached device memory that its peers map through hipIpc memory handles
 * (one node, xGMI or - for two rank processes sharing one GPU - the same HBM).  The bucket is cut into chunks of 1024
 * floats; chunk c is OWNED by rank (c mod nranks).  One launch per rank does the whole exchange, workgroup c = chunk c:
 *   push     a non-owner stores its 4 KiB of chunk c into slot [its rank] of the owner's window (write-through stores,
 *            drained, then ONE system-scope flag store) - remote accesses are stores only, every load is local
 *   reduce   the owner waits for the nranks-1 flags of the chunk, adds the contributions IN RANK ORDER, stores the
 *            sum into every peer's window and raises the peers' "reduced" flag of the chunk
 *   consume  a non-owner waits for that flag and reads the sum from its own window
 * so every rank ends with the SAME bits (the owner's sum), the same in every run.  A flag carries the number of exchanges
 * its chunk has been through; each workgroup keeps that count for its chunk in device memory, so a launch recorded in a
 * hipGraph replays correctly, no flag is ever reset and no workgroup waits for another one of its own launch.  A wait gives up
 * after LG_P2P_TIMEOUT_S seconds (a lost peer): the device status flag is raised and the next synchronising call of lghip.h
 * returns LG_ECOMM; the communicator stays failed from then on (lg_p2p_state).
 *
 * lg_p2p_adam_multi_dev_f32 is lg_adam_multi_dev_f32 (lghip.h) with this exchange in front of the update, in the SAME
 * launch: the data-parallel training step keeps the launch count of the single-GPU step.
 *
 * One rank per GPU is the deployment.  Rank processes that share ONE GPU (how this path is tested on one-GPU machines:
 * hipIpc handles are per process, not per device) must each be confined to CUs of their own - LG_CU_MASK=<k>/<n> in the
 * environment before lg_init - or a rank's waiting workgroups can keep its peer's kernels off the device.
 *
 * Rendezvous: lg_p2p_export gives 64 bytes that the launcher's rendezvous files carry to the peers (like the RCCL id of
 * lghip_comm.h); lg_p2p_connect takes all ranks' handles.  All functions return 0 or a negative LG_E* code (lghip.h)
 * with the message in lg_last_error(); launches go to the compute stream (lg_stream()).
 */
#ifndef LGHIP_P2P_H
#define LGHIP_P2P_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_P2P_HANDLE_BYTES 64
#define LG_P2P_MAX_RANKS    8
#define LG_P2P_TIMEOUT_S    20

typedef enum { LG_P2P_SUM = 0, LG_P2P_MAX = 1 } lg_p2p_op_t;

/* allocate this rank's window for buckets of up to `capacity_floats` floats and give its hipIpc handle */
int lg_p2p_export(int rank, int nranks, int64_t capacity_floats, char handle[LG_P2P_HANDLE_BYTES]);
/* map the windows of all peers: `handles` = nranks x LG_P2P_HANDLE_BYTES, ordered by rank (the own entry is ignored) */
int lg_p2p_connect(const char* handles);
int lg_p2p_rank(int* rank, int* nranks, int64_t* capacity_floats);
/* failed: 1 once a wait of this communicator gave up and a synchronising call reported it (LG_ECOMM) - from then on every
 * lg_p2p_allreduce_f32 / lg_p2p_adam_multi_dev_f32 returns LG_ECOMM without launching, and launches that were enqueued (or
 * recorded in a hipGraph) before are reported again at the next synchronising call: an exchange that did not happen is never
 * taken for one that did.  The way out is lg_p2p_disconnect + lg_p2p_free (no barrier with the lost peer) and a new window.
 * memory_kind: how the window was allocated ("uncached", "fine-grained", "hipMalloc"). */
int lg_p2p_state(int* failed, char memory_kind[32]);
/* tests only: pretend every chunk has been through `epoch` exchanges (epochs count modulo 2^32; the wrap is tested by
 * seeding them just below it).  Same value on every rank, no launch in flight, host barrier between the ranks afterwards. */
int lg_p2p_debug_seed_epochs(int epoch);
/* in-place all-reduce of buf[0..n) over the ranks; n may exceed the capacity (then several launches).  Collective:
 * every rank must issue the same sequence of lg_p2p_* launches with the same sizes. */
int lg_p2p_allreduce_f32(float* buf, int64_t n, int op);
/* lg_adam_multi_dev_f32 with the all-reduce (SUM) of g in front of the update, in the same launch; g holds the summed
 * gradient afterwards.  The launch also advances the optimizer's step number, without any cross-workgroup hand-off:
 * `step` points to 2 + step_slots int64, all holding the number of steps done so far (zeros for a new optimizer);
 * step[2 + c] is the private copy of the workgroup of chunk c, step[0] the copy the caller reads.  step_slots >= the
 * number of 1024-float chunks of the bucket, each segment rounded up: sum over segments of ceil(length / 1024). */
int lg_p2p_adam_multi_dev_f32(float* p, float* g, float* m, float* v, int nseg, const int64_t* offsets,
                              double lr, double b1, double b2, double eps, int64_t* step, int64_t step_slots, double gscale,
                              int belief);
/* unmap the peers' windows (call on every rank, then synchronise the ranks, then lg_p2p_free) */
int lg_p2p_disconnect(void);
int lg_p2p_free(void);

#ifdef __cplusplus
}
#endif
#endif /* LGHIP_P2P_H */
