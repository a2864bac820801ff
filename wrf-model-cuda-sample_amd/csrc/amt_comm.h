// amt_comm.h -- the halo-exchange engine the j-slab stepper (amt_slab.hip) and the i x j grid stepper (amt_grid.hip)
// share (not installed).  An exchange is a fixed list of contiguous SEND segments (device pointer, bytes, destination
// rank) and RECEIVE segments (device pointer, bytes, source rank) of one rank, set once at creation; between two ranks the
// k-th segment one sends to the other is the k-th segment the other receives from it.  Two transports carry it:
//   RCCL  one ncclSend / ncclRecv group per exchange (north_star's transport: xGMI between the GPUs of a node).  RCCL's
//         send/recv kernel holds ~31 workgroups while it waits for the wire, and it refuses two ranks on one device.
//   IPC   no RCCL: the ranks trade hipIpcMemHandles of their send segments once, through a POSIX shared-memory block
//         that also holds a mailbox of sequence numbers; per exchange the receiver waits for the sender's "rows final"
//         number (a one-wave kernel polling the mailbox), PULLS the rows with the copy engine (hipMemcpyAsync from the
//         peer mapping: SDMA over xGMI between GPUs, no compute unit) and posts "pulled"; the sender's sweep ends when
//         its rows have been pulled.  Works between processes that share ONE device (profiles/r05_ipc_probe.hip).
#pragma once
#include "amt_internal.h"

struct AmtSeg {
    void *ptr;
    size_t bytes;
    int peer;
};

enum { AMT_XCHG_RCCL = 0, AMT_XCHG_IPC = 1 };

struct AmtExchange;

// Collective over the `world` ranks.  `unique_id`: the AMT_UNIQUE_ID_BYTES every rank got from rank 0 (amt_comm_unique_id).
// self_loop: world == 1 and every peer is this rank (the one-GPU test mode of both steppers).
int amt_exchange_create(AmtExchange **out, int transport, int rank, int world, const void *unique_id, int device,
                        const AmtSeg *sends, int nsend, const AmtSeg *recvs, int nrecv, bool self_loop);
int amt_exchange_destroy(AmtExchange *x);
// Phase A on `stream`: when it has run, every receive segment holds the sender's current rows.
// alone: nothing else of this rank runs meanwhile (no-overlap schedules): the IPC pull may take the whole chip
int amt_exchange_enqueue(AmtExchange *x, hipStream_t stream, bool alone = false);
// The same phase for the IPC transport with the wait on the HOST instead of in a kernel (no compute unit is held while the
// neighbour is late): post on the stream that made the rows final, poll on the calling thread, pull on `stream`.
int amt_exchange_enqueue_post(AmtExchange *x, hipStream_t stream);
int amt_exchange_host_wait(AmtExchange *x);
int amt_exchange_enqueue_pull(AmtExchange *x, hipStream_t stream);
// Phase B on `stream` (IPC; nothing for RCCL, whose sends complete inside the group): when it has run, every destination
// has pulled this exchange's rows -- the send segments may be overwritten.
int amt_exchange_enqueue_release(AmtExchange *x, hipStream_t stream);
// Host: has a device-side wait of this rank given up (a neighbour that never posted)?  AMT_ERR_COMM then.
int amt_exchange_check(AmtExchange *x);
// rank and size as the transport itself reports them (ncclCommUserRank / ncclCommCount; the ranks attached to the block)
int amt_exchange_info(const AmtExchange *x, int *rank, int *world);
// host-side max over the ranks of *x, also a barrier; `stream` (drained by the caller) carries the RCCL all-reduce
int amt_exchange_max(AmtExchange *x, double *v, hipStream_t stream);
int amt_exchange_transport(const AmtExchange *x);
void amt_exchange_bytes(const AmtExchange *x, size_t *sent, size_t *received);      // per exchange, this rank
void amt_exchange_set_skew_us(AmtExchange *x, int microseconds);    // test hook: phase A completes this late
bool amt_exchange_owns_skew(const AmtExchange *x);
const char *amt_exchange_pull_mode(const AmtExchange *x);             // "fused kernel" / "copy engine" / "" (RCCL)
bool amt_exchange_active(const AmtExchange *x);       // false: no segment at all (a world of one without loopback)
