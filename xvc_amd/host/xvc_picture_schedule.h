// xvc_picture_schedule.h -- picture-level parallelism the way the reference
// does it (second row): the pictures of a sub-GOP that do not
// depend on one another are coded at the same time, by worker threads in the
// reference (xvc_enc_lib/thread_encoder.cc:61-159), by GPUs (ranks) and by the
// picture slots of one GPU here.  Pure host logic, no device calls:
//
//   * SubGop: coding order, POC and temporal layer of a sub-GOP
//     (SegmentHeader::CalcPocFromDoc / CalcDocFromPoc / CalcTidFromDoc,
//     xvc_common_lib/segment_header.cc:135-175; the dyadic rows - lengths 1, 2,
//     4, ..., 64 - of its tables as arithmetic);
//   * ReferenceLists: the L0 / L1 lists and with them the dependencies of a
//     picture (ReferenceListSorter::Prepare with FillLowerPoc / FillHigherPoc,
//     xvc_common_lib/reference_list_sorter.h:46-164; random-access
//     configuration, one segment);
//   * ThreadEncoderSchedule: the worker pool's policy - a free worker takes,
//     among the queued pictures whose dependencies are all finished, the one
//     with the lowest temporal layer, first queued first on a tie
//     (ThreadEncoder::WorkerMain, thread_encoder.cc:99-159) - played forward
//     with per-layer picture durations, which fixes for every picture its
//     worker and start time, and for every reconstructed reference picture the
//     ranks that need a copy;
//   * the TIMELINE: encodes and reference transfers in one global order that
//     every rank walks (each acting only on the entries that name it).  A
//     transfer is numbered by the time its picture finishes, so everything a
//     transfer waits for has a lower number: walking the same order on every
//     rank pairs sends with receives and cannot deadlock.
#ifndef XVC_AMD_HOST_XVC_PICTURE_SCHEDULE_H_
#define XVC_AMD_HOST_XVC_PICTURE_SCHEDULE_H_

#include <cstdint>
#include <vector>

extern "C" {

// One picture of the sequence, in coding order (index = position in the
// vector the schedule was built from).
typedef struct xvc_sched_picture {
  int32_t poc, doc, tid;
  int32_t intra;           // no references (the first picture of the segment)
  int32_t num_ref[2];
  int32_t ref_poc[2][5];   // L0 / L1 as the reference builds them, -1 = unused
  int32_t is_reference;    // some later picture lists it
  int32_t worker;          // ThreadEncoder worker that codes it
  int32_t rank, slot;      // worker -> (rank, picture slot of that rank)
  int32_t start, finish;   // schedule time units
} xvc_sched_picture;

enum { XVC_SCHED_ENCODE = 0, XVC_SCHED_TRANSFER = 1 };

// One entry of the timeline.
typedef struct xvc_sched_op {
  int32_t kind;            // XVC_SCHED_ENCODE / XVC_SCHED_TRANSFER
  int32_t picture;         // index into the picture array
  int32_t src_rank;        // ENCODE: the rank that codes it; TRANSFER: the owner
  int32_t dst_rank;        // TRANSFER: the rank that receives a copy (ENCODE: -1)
  int32_t time;
} xvc_sched_op;

typedef struct xvc_schedule xvc_schedule;

// num_pictures pictures (POC 0 .. num_pictures-1, POC 0 intra) in sub-GOPs of
// sub_gop_length (a power of two <= 64), num_ref_pics per list (<= 5), played
// on `ranks` ranks with `slots_per_rank` pictures in flight each.  layer_cost[t]
// = duration of a picture of temporal layer t in schedule time units (nullptr:
// 1 each).  Returns nullptr on invalid arguments.
xvc_schedule *xvc_schedule_create(int num_pictures, int sub_gop_length, int num_ref_pics,
                                  int ranks, int slots_per_rank, const int32_t *layer_cost,
                                  int num_layers);
void xvc_schedule_destroy(xvc_schedule *s);
int xvc_schedule_num_pictures(const xvc_schedule *s);
const xvc_sched_picture *xvc_schedule_pictures(const xvc_schedule *s);
int xvc_schedule_num_ops(const xvc_schedule *s);
const xvc_sched_op *xvc_schedule_ops(const xvc_schedule *s);
int xvc_schedule_makespan(const xvc_schedule *s);
// Pictures in flight at most (see PictureSchedule::window): a ring of
// window + 2 * sub_gop_length picture buffers per rank is enough.
int xvc_schedule_window(const xvc_schedule *s);

// SegmentHeader::CalcDocFromPoc / CalcPocFromDoc / CalcTidFromDoc for dyadic
// sub-GOP lengths, with sub_gop_start_poc = the start of the picture's own
// sub-GOP as the encoder passes it (encoder.cc:97); -1 for a length this file
// does not restate.
int xvc_sched_doc_from_poc(int poc, int sub_gop_length);
int xvc_sched_poc_from_doc(int doc, int sub_gop_length);
int xvc_sched_tid_from_doc(int doc, int sub_gop_length);

// What a rank does for one entry (callbacks of xvc_schedule_run).  All return
// 0 on success; a non-zero value stops the walk and is returned.
typedef struct xvc_sched_callbacks {
  // code picture `p` (refs are pictures[...] indices by POC lookup done by the
  // callee through p->ref_poc) in picture slot p->slot
  int (*encode)(void *user, const xvc_sched_picture *p, int picture_index);
  // ship / receive the padded reconstruction of picture `p`
  int (*send)(void *user, const xvc_sched_picture *p, int picture_index, int dst_rank);
  int (*recv)(void *user, const xvc_sched_picture *p, int picture_index, int src_rank);
} xvc_sched_callbacks;

// Walks the timeline as rank `rank`: every ENCODE whose src_rank is `rank`,
// every TRANSFER that names it as source (send) or destination (recv), in
// timeline order.
int xvc_schedule_run(const xvc_schedule *s, int rank, const xvc_sched_callbacks *cb, void *user);
// The same over timeline entries [first_op, end_op) only (a warm-up part and a
// timed part of one sequence).
int xvc_schedule_run_range(const xvc_schedule *s, int rank, const xvc_sched_callbacks *cb,
                           void *user, int first_op, int end_op);

}  // extern "C"

namespace xvc_gpu {

struct SubGop {
  // dyadic sub-GOP of `length` pictures: doc 1 is the picture at POC `length`
  // (layer 0), layer t >= 1 holds the odd multiples of length / 2^t in
  // ascending POC order
  static bool Supported(int length);
  static int DocFromPoc(int poc, int length);
  static int PocFromDoc(int doc, int length);
  static int TidFromDoc(int doc, int length);
};

class PictureSchedule {
 public:
  PictureSchedule(int num_pictures, int sub_gop_length, int num_ref_pics, int ranks,
                  int slots_per_rank, const std::vector<int> &layer_cost);
  const std::vector<xvc_sched_picture> &pictures() const { return pics_; }
  const std::vector<xvc_sched_op> &ops() const { return ops_; }
  int makespan() const { return makespan_; }
  // pictures that can be queued at once (the reference's pic_buffering_num_ with
  // one extra sub-GOP per additional worker, encoder.cc:245-260): no picture
  // starts while one more than `window` places before it is unfinished
  int window() const { return window_; }
  int Run(int rank, const xvc_sched_callbacks &cb, void *user, int first_op = 0,
          int end_op = -1) const;

 private:
  void BuildSequence(int num_pictures, int sub_gop_length);
  void BuildReferenceLists(int num_ref_pics);
  void Play(int ranks, int slots_per_rank, const std::vector<int> &layer_cost);
  void BuildTimeline();
  int IndexOfPoc(int poc) const;

  std::vector<xvc_sched_picture> pics_;
  std::vector<std::vector<int>> deps_;  // picture index -> picture indices it lists
  std::vector<xvc_sched_op> ops_;
  int sub_gop_length_, num_ref_pics_, window_, makespan_;
};

}  // namespace xvc_gpu
#endif  // XVC_AMD_HOST_XVC_PICTURE_SCHEDULE_H_
