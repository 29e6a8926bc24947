/* lghip_comm.h - C ABI of liblghip_comm.so: data-parallel collectives over
 * RCCL/xGMI for the HipTensor backend.  Separate shared object so that the
 * single-GPU path never loads RCCL.
 *
 * The reference has no distributed code at all; this is the
 * MI355X-native addition for BASELINE config #4: one process per GPU, one
 * fp32 sum all-reduce per training step over a single flat gradient bucket,
 * with no host synchronisation.  The exchange runs on the library's own
 * communication stream, forked from the compute stream (lg_stream()) as soon
 * as the last parameter gradient has been enqueued and joined before the
 * optimizer (lg_comm_fork / lg_comm_allreduce_forked_f32 / lg_comm_join), so
 * it overlaps the rest of backward; during lg_graph_begin .. lg_graph_end the
 * same three calls are captured as a parallel branch of the hipGraph.
 *
 * Rendezvous: rank 0 calls lg_comm_get_unique_id and hands the 128 bytes to
 * the other ranks out of band (the launcher's file/pipe); no network service.
 * All functions return 0 or a negative LG_E* code (lghip.h); the message is
 * available from lg_comm_last_error().
 */
#ifndef LGHIP_COMM_H
#define LGHIP_COMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LG_COMM_ID_BYTES 128

typedef enum { LG_COMM_SUM = 0, LG_COMM_MAX = 1 } lg_comm_op_t;

const char* lg_comm_last_error(void);
int lg_comm_get_unique_id(char id[LG_COMM_ID_BYTES]);
int lg_comm_init(int rank, int nranks, const char id[LG_COMM_ID_BYTES]);   /* lg_init must have been called */
int lg_comm_rank(int* rank, int* nranks);
/* first collective of a new communicator: all-reduce of 1.0 on the COMMUNICATION stream, awaited by polling an event for at
 * most `timeout_s` seconds (LG_ECOMM after that, or when the sum is not nranks) - the compute stream has nothing of it queued,
 * so a job can go on with another form of the exchange (lghip_p2p.h) when a peer fell out during lg_comm_init */
int lg_comm_selftest(double timeout_s);
int lg_comm_allreduce_f32(float* buf, int64_t n, int op);                  /* in place, on the compute stream */
/* overlap: fork = the communication stream waits for everything enqueued so far on the compute stream;
 * allreduce_forked = the collective, in place, on the communication stream; join = the compute stream waits
 * for everything enqueued so far on the communication stream.  No host synchronisation in any of them. */
int lg_comm_fork(void);
int lg_comm_allreduce_forked_f32(float* buf, int64_t n, int op);
int lg_comm_join(void);
int lg_comm_broadcast_f32(float* buf, int64_t n, int root);                /* in place, stream-ordered */
int lg_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* LGHIP_COMM_H */
