/* lcd_p2p.h -- C-ABI of the one-shot direct peer-to-peer exchanges of the sharded frame (liblcd_p2p.so), the hand-rolled counterpart of the
 * two RCCL calls include/lcd_shard.h makes per frame (SURVEY.md section 5 "Distributed communication backend" and section 8e "Collective
 * choice": messages of 8 KB ... 8 MB are latency-bound on a ring of 2 (p - 1) steps; with 8 fully connected GPUs every peer can be written
 * at once over its own xGMI link).  The reference has no multi-GPU path ("Collectives: none"); what these exchanges carry is
 * what VWDictionary::addNewWords' nearest-neighbour results (VWDictionary.cpp:1015-1086) and Memory::computeLikelihood's sums
 * (Memory.cpp:2215-2291) become when the vocabulary is sharded by word id.
 *
 * One lcd_p2p per rank (one process per GPU).  Every rank owns an ARENA of uncached (fine-grained) device memory that its peers map
 * through hipIpc handles: flags, a double-buffered mailbox for the all-gather, a staging area for the all-reduce.  Kernels write peers'
 * arenas directly and announce it with a release store to a flag in the peer's arena; the receiving kernel polls that flag (acquire),
 * bounded by a wall-clock timeout -- an exchange whose peer never arrives ends with a status bit (lcd_p2p_status), not with a hung GPU.
 *     all-gather: ONE kernel -- workgroup group p pushes this rank's block into peer p's mailbox[epoch & 1][rank], raises peer p's flag,
 *                 waits for peer p's block in its own mailbox and copies it out.
 *     all-reduce: three kernels -- stage (buffer -> own arena, flag A to every peer) | reduce-scatter + all-gather in place (rank r waits
 *                 for every A, sums slice r of every arena in RANK ORDER and writes the sum back into slice r of every arena, flag B) |
 *                 collect (waits for every B, own arena -> buffer).  Each phase moves count / world elements per link, all links at once.
 *                 The wire is the buffer's 64-bit integers (bit-identical to any other algorithm) or, with LCD_P2P_WIRE_F32, 32-bit
 *                 floats converted while staging and back while collecting (half the bytes; |error| <= world * 2^-24 relative per sum).
 * Setup: every rank calls lcd_p2p_create with the SAME capacities, exports LCD_P2P_HANDLE_BYTES, the caller carries the world's exports
 * to every rank by whatever it has (MPI, torch.distributed, a file), lcd_p2p_connect maps them.  Ranks of one process (several GPUs
 * driven by one host thread each) connect through the raw pointers in the export instead of hipIpc.
 * Lifetime: a rank destroys its lcd_p2p only after EVERY rank has completed its exchanges (a barrier of the caller's: a peer's kernel may still be
 * writing this rank's arena).  After a time-out the result of that exchange is undefined on the rank that gave up (its buffer holds whatever lay in
 * the arena); the ranks are in step again as soon as every rank has made the call -- epochs are counted per call, not per success.
 * Ordering contract: every rank makes the same calls in the same order; all-gathers are enqueued on ONE stream per rank; all-reduces are
 * ordered among themselves (the driver of lcd_shard.h does both) -- an all-gather may run beside an all-reduce (separate flags and memory).
 * extern "C", plain pointers and sizes, the status codes of lcd.h; nothing throws across the boundary. */
#ifndef LCD_P2P_H_
#define LCD_P2P_H_

#include "lcd_shard.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LCD_P2P_ABI_VERSION 1
#define LCD_P2P_MAX_WORLD 16
#define LCD_P2P_HANDLE_BYTES 128

typedef struct lcd_p2p lcd_p2p;

enum lcd_p2p_wire {
    LCD_P2P_WIRE_I64 = 0,      /* the all-reduce moves the buffer's 64-bit integers: order-free, bit-identical to RCCL's result */
    LCD_P2P_WIRE_F32 = 1       /* ... 32-bit floats ("4 B x N_sig"), summed in rank order: deterministic for a given world */
};

/* status bits of lcd_p2p_status (sticky until lcd_p2p_clear_status) */
#define LCD_P2P_TIMEOUT_GATHER 1u   /* a peer's all-gather block did not arrive in time */
#define LCD_P2P_TIMEOUT_STAGE 2u    /* a peer's staged all-reduce operand (flag A) did not arrive in time */
#define LCD_P2P_TIMEOUT_REDUCE 4u   /* a peer's reduced slice (flag B) did not arrive in time */

/* rank `rank` of `world` (1..LCD_P2P_MAX_WORLD) on the calling thread's current device.  gather_bytes_per_rank_max: the largest block one
 * rank contributes to an all-gather (q x 2 records of 16 bytes for lcd_shard.h); reduce_count_max: the largest element count of an
 * all-reduce (signature slots + 1).  Same values on every rank. */
int lcd_p2p_create(int rank, int world, size_t gather_bytes_per_rank_max, size_t reduce_count_max, lcd_p2p** out);
/* this rank's export: hipIpc handle of the arena + what lcd_p2p_connect checks (capacities, world, pid, pointer) */
int lcd_p2p_export(lcd_p2p* p, unsigned char out[LCD_P2P_HANDLE_BYTES]);
/* all_exports: world x LCD_P2P_HANDLE_BYTES, rank-major (this rank's own entry included).  Maps every peer's arena. */
int lcd_p2p_connect(lcd_p2p* p, const unsigned char* all_exports);
void lcd_p2p_destroy(lcd_p2p* p);
const char* lcd_p2p_last_error(const lcd_p2p* p);

int lcd_p2p_set_wire(lcd_p2p* p, int wire);                 /* enum lcd_p2p_wire; same value on every rank, between exchanges */
/* on = 1: every wave publishes with the compiler's system-scope release fence and the flags are release stores (an L2 write-back per wave:
 * 2 - 4 x slower, profiles/r06_p2p_exchange.txt) instead of "wait for the stores' acknowledgement, then a relaxed flag", which rests on the
 * arenas being mapped uncached on BOTH sides.  A bring-up switch for a fabric this library has not run on (it has only run between processes
 * of one GPU): if results differ there, turn it on first.  Same value on every rank; default 0. */
int lcd_p2p_set_conservative_fences(lcd_p2p* p, int on);
int lcd_p2p_set_timeout_ms(lcd_p2p* p, int64_t ms);         /* how long a kernel polls a flag before it gives up (default 10 000) */
uint32_t lcd_p2p_status(const lcd_p2p* p);                  /* LCD_P2P_TIMEOUT_* bits raised by kernels that have completed */
void lcd_p2p_clear_status(lcd_p2p* p);

/* recv[r * bytes_per_rank ..) = rank r's send[0 .. bytes_per_rank) for every r; bytes_per_rank a multiple of 16, device buffers aligned to
 * 16 bytes.  Enqueued on `stream` (hipStream_t); nothing is synchronised. */
int lcd_p2p_all_gather(lcd_p2p* p, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);
/* buf[i] = sum over the ranks of buf[i], in place, `count` 64-bit integers in a buffer aligned to 16 bytes (wire as lcd_p2p_set_wire says) */
int lcd_p2p_all_reduce_sum_i64(lcd_p2p* p, void* d_buf, size_t count, void* stream);

/* the two exchanges as the callbacks lcd_shard_comm_create_transport takes (out->user = p: p must outlive the communicator) */
int lcd_p2p_transport(lcd_p2p* p, lcd_shard_transport* out);

#ifdef __cplusplus
}
#endif
#endif /* LCD_P2P_H_ */
