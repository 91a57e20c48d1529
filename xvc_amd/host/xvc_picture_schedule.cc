// xvc_picture_schedule.cc -- see xvc_picture_schedule.h.
#include "xvc_picture_schedule.h"

#include <algorithm>
#include <new>

namespace xvc_gpu {

namespace {
int Log2(int v) {
  int l = 0;
  while ((1 << (l + 1)) <= v) l++;
  return l;
}
int Ctz(int v) {
  int c = 0;
  while (!(v & 1)) {
    v >>= 1;
    c++;
  }
  return c;
}
}  // namespace

bool SubGop::Supported(int length) {
  return length >= 1 && length <= 64 && (length & (length - 1)) == 0;
}

// Within a sub-GOP (positions 1..length): position `length` is coded first
// (layer 0); layer t >= 1 = the odd multiples of length / 2^t, ascending.
int SubGop::DocFromPoc(int poc, int length) {
  if (poc < 1) return 0;
  const int base = (poc - 1) / length * length, rem = poc - base;  // rem in 1..length
  if (rem == length) return base + 1;
  const int k = Log2(length), t = k - Ctz(rem);
  const int idx = ((rem >> (k - t)) - 1) >> 1;
  return base + (1 << (t - 1)) + 1 + idx;
}

int SubGop::PocFromDoc(int doc, int length) {
  if (doc < 1) return 0;
  const int base = (doc - 1) / length * length, rem = doc - base;
  if (rem == 1) return base + length;
  const int t = Log2(rem - 1) + 1;  // layer: docs 2^(t-1)+1 .. 2^t
  const int idx = rem - (1 << (t - 1)) - 1;
  return base + (2 * idx + 1) * (length >> t);
}

int SubGop::TidFromDoc(int doc, int length) {
  if (doc < 1) return 0;
  const int rem = (doc - 1) % length + 1;
  return rem == 1 ? 0 : Log2(rem - 1) + 1;
}

PictureSchedule::PictureSchedule(int num_pictures, int sub_gop_length, int num_ref_pics,
                                 int ranks, int slots_per_rank,
                                 const std::vector<int> &layer_cost)
    : sub_gop_length_(sub_gop_length), num_ref_pics_(num_ref_pics), window_(0), makespan_(0) {
  BuildSequence(num_pictures, sub_gop_length);
  BuildReferenceLists(num_ref_pics);
  Play(ranks, slots_per_rank, layer_cost);
  BuildTimeline();
}

int PictureSchedule::IndexOfPoc(int poc) const {
  for (size_t i = 0; i < pics_.size(); i++)
    if (pics_[i].poc == poc) return static_cast<int>(i);
  return -1;
}

// Coding order: ascending doc over the pictures that exist (a last, shorter
// sub-GOP keeps the docs of the full one: encoder.cc:169-185 codes "the picture
// with the next doc").
void PictureSchedule::BuildSequence(int num_pictures, int length) {
  sub_gop_length_ = length;
  pics_.clear();
  for (int poc = 0; poc < num_pictures; poc++) {
    xvc_sched_picture p = xvc_sched_picture();
    p.poc = poc;
    p.doc = SubGop::DocFromPoc(poc, length);
    p.tid = SubGop::TidFromDoc(p.doc, length);
    p.intra = poc == 0;
    for (int l = 0; l < 2; l++)
      for (int k = 0; k < 5; k++) p.ref_poc[l][k] = -1;
    p.worker = p.rank = p.slot = -1;
    pics_.push_back(p);
  }
  std::sort(pics_.begin(), pics_.end(),
            [](const xvc_sched_picture &a, const xvc_sched_picture &b) { return a.doc < b.doc; });
}

// ReferenceListSorter::Prepare, random access (reference_list_sorter.h:72-84):
//   L0 = FillLowerPoc, and FillHigherPoc only if that found nothing;
//   L1 = FillHigherPoc, then FillLowerPoc from where it stopped.
// Fill*Poc walk away from the current picture, each step taking the closest
// picture on that side whose layer is below the last one taken (or is 0).
// Candidates: the pictures coded before (lower doc) that the reference's buffer
// of sub_gop_length + num_ref_pics + 1 pictures (encoder.cc:251-257) still
// holds - modelled as everything from the previous sub-GOP's anchor on.
void PictureSchedule::BuildReferenceLists(int num_ref_pics) {
  const int L = sub_gop_length_;
  deps_.assign(pics_.size(), std::vector<int>());
  for (size_t i = 0; i < pics_.size(); i++) {
    xvc_sched_picture &p = pics_[i];
    if (p.intra) continue;
    const int anchor = (p.poc - 1) / L * L;
    const int oldest = anchor - L;
    auto fill = [&](int list, int start, bool lower) {
      int last_poc = p.poc, last_tid = p.tid, idx = start;
      while (idx < num_ref_pics) {
        int best = -1;
        for (size_t j = 0; j < i; j++) {
          const xvc_sched_picture &c = pics_[j];
          if (c.poc < oldest) continue;
          if (!(c.tid < last_tid || c.tid == 0)) continue;
          if (lower) {
            bool listed = false;  // HasRefPoc (FillLowerPoc only, :101)
            for (int k = 0; k < idx; k++) listed |= p.ref_poc[list][k] == c.poc;
            if (listed || c.poc >= last_poc) continue;
            if (best < 0 || c.poc > pics_[best].poc) best = static_cast<int>(j);
          } else {
            if (c.poc <= last_poc) continue;
            if (best < 0 || c.poc < pics_[best].poc) best = static_cast<int>(j);
          }
        }
        if (best < 0) break;
        last_poc = pics_[best].poc;
        last_tid = pics_[best].tid;
        p.ref_poc[list][idx++] = last_poc;
        if (std::find(deps_[i].begin(), deps_[i].end(), best) == deps_[i].end())
          deps_[i].push_back(best);
        pics_[best].is_reference = 1;
      }
      return idx;
    };
    int n0 = fill(0, 0, true);
    if (n0 == 0) n0 = fill(0, n0, false);
    int n1 = fill(1, 0, false);
    n1 = fill(1, n1, true);
    p.num_ref[0] = n0;
    p.num_ref[1] = n1;
  }
}

// ThreadEncoder::WorkerMain played forward.  All pictures are queued in coding
// order; whenever a worker is free it takes the ready picture (all
// dependencies finished) with the lowest tid, the earliest queued on a tie;
// workers are served in index order.  Worker w is picture slot w / ranks of
// rank w % ranks: pictures that run at the same time spread over the ranks
// before they share one.
void PictureSchedule::Play(int ranks, int slots_per_rank, const std::vector<int> &layer_cost) {
  const int workers = ranks * slots_per_rank;
  const int n = static_cast<int>(pics_.size());
  window_ = num_ref_pics_ + sub_gop_length_ * workers + 1;
  std::vector<int> busy_until(workers, 0);
  std::vector<char> taken(n, 0);
  int done = 0, now = 0;
  auto cost = [&](int tid) {
    if (layer_cost.empty()) return 1;
    const int c = layer_cost[std::min<size_t>(tid, layer_cost.size() - 1)];
    return c > 0 ? c : 1;
  };
  while (done < n) {
    bool progressed = false;
    for (int w = 0; w < workers; w++) {
      if (busy_until[w] > now) continue;
      int best = -1;
      int first_open = 0;  // the oldest picture not finished yet
      while (first_open < n && taken[first_open] && pics_[first_open].finish <= now) first_open++;
      for (int i = first_open; i < n && i < first_open + window_; i++) {
        if (taken[i]) continue;
        bool ready = true;
        for (int d : deps_[i]) ready &= taken[d] && pics_[d].finish <= now;
        if (!ready) continue;
        if (best < 0 || pics_[i].tid < pics_[best].tid) best = i;
      }
      if (best < 0) break;  // nothing ready: no later worker finds anything either
      xvc_sched_picture &p = pics_[best];
      taken[best] = 1;
      p.worker = w;
      p.rank = w % ranks;
      p.slot = w / ranks;
      p.start = now;
      p.finish = now + cost(p.tid);
      busy_until[w] = p.finish;
      makespan_ = std::max(makespan_, p.finish);
      done++;
      progressed = true;
    }
    if (done == n) break;
    // advance to the next completion
    int next = -1;
    for (int w = 0; w < workers; w++)
      if (busy_until[w] > now && (next < 0 || busy_until[w] < next)) next = busy_until[w];
    if (next < 0) {
      if (!progressed) break;  // cannot happen: dependencies point to earlier docs
      continue;
    }
    now = next;
  }
}

// Encodes at their start time, transfers at their picture's finish time; at
// equal times transfers first (what a picture starting then may need), then
// coding order.  Every rank walks this one list.
void PictureSchedule::BuildTimeline() {
  ops_.clear();
  const int n = static_cast<int>(pics_.size());
  for (int i = 0; i < n; i++) {
    xvc_sched_op e = {XVC_SCHED_ENCODE, i, pics_[i].rank, -1, pics_[i].start};
    ops_.push_back(e);
    std::vector<int> consumers;
    for (int j = 0; j < n; j++)
      if (std::find(deps_[j].begin(), deps_[j].end(), i) != deps_[j].end() &&
          pics_[j].rank != pics_[i].rank &&
          std::find(consumers.begin(), consumers.end(), pics_[j].rank) == consumers.end())
        consumers.push_back(pics_[j].rank);
    std::sort(consumers.begin(), consumers.end());
    for (int r : consumers) {
      xvc_sched_op t = {XVC_SCHED_TRANSFER, i, pics_[i].rank, r, pics_[i].finish};
      ops_.push_back(t);
    }
  }
  std::stable_sort(ops_.begin(), ops_.end(), [this](const xvc_sched_op &a, const xvc_sched_op &b) {
    if (a.time != b.time) return a.time < b.time;
    if (a.kind != b.kind) return a.kind == XVC_SCHED_TRANSFER;
    if (a.picture != b.picture) return a.picture < b.picture;
    return a.dst_rank < b.dst_rank;
  });
}

int PictureSchedule::Run(int rank, const xvc_sched_callbacks &cb, void *user, int first_op,
                         int end_op) const {
  const int n = static_cast<int>(ops_.size());
  if (end_op < 0 || end_op > n) end_op = n;
  for (int k = first_op < 0 ? 0 : first_op; k < end_op; k++) {
    const xvc_sched_op &op = ops_[k];
    const xvc_sched_picture &p = pics_[op.picture];
    int st = 0;
    if (op.kind == XVC_SCHED_ENCODE) {
      if (op.src_rank == rank && cb.encode) st = cb.encode(user, &p, op.picture);
    } else if (op.src_rank == rank) {
      if (cb.send) st = cb.send(user, &p, op.picture, op.dst_rank);
    } else if (op.dst_rank == rank) {
      if (cb.recv) st = cb.recv(user, &p, op.picture, op.src_rank);
    }
    if (st) return st;
  }
  return 0;
}

}  // namespace xvc_gpu

struct xvc_schedule {
  xvc_gpu::PictureSchedule s;
  xvc_schedule(int n, int l, int r, int ranks, int slots, const std::vector<int> &c)
      : s(n, l, r, ranks, slots, c) {}
};

extern "C" {

xvc_schedule *xvc_schedule_create(int num_pictures, int sub_gop_length, int num_ref_pics,
                                  int ranks, int slots_per_rank, const int32_t *layer_cost,
                                  int num_layers) {
  if (num_pictures < 1 || !xvc_gpu::SubGop::Supported(sub_gop_length) || num_ref_pics < 1 ||
      num_ref_pics > 5 || ranks < 1 || slots_per_rank < 1 || num_layers < 0)
    return nullptr;
  std::vector<int> cost;
  if (layer_cost) cost.assign(layer_cost, layer_cost + num_layers);
  return new (std::nothrow)
      xvc_schedule(num_pictures, sub_gop_length, num_ref_pics, ranks, slots_per_rank, cost);
}
void xvc_schedule_destroy(xvc_schedule *s) { delete s; }
int xvc_schedule_num_pictures(const xvc_schedule *s) {
  return s ? static_cast<int>(s->s.pictures().size()) : 0;
}
const xvc_sched_picture *xvc_schedule_pictures(const xvc_schedule *s) {
  return s ? s->s.pictures().data() : nullptr;
}
int xvc_schedule_num_ops(const xvc_schedule *s) {
  return s ? static_cast<int>(s->s.ops().size()) : 0;
}
const xvc_sched_op *xvc_schedule_ops(const xvc_schedule *s) {
  return s ? s->s.ops().data() : nullptr;
}
int xvc_schedule_makespan(const xvc_schedule *s) { return s ? s->s.makespan() : 0; }
int xvc_schedule_window(const xvc_schedule *s) { return s ? s->s.window() : 0; }

int xvc_sched_doc_from_poc(int poc, int l) {
  return xvc_gpu::SubGop::Supported(l) ? xvc_gpu::SubGop::DocFromPoc(poc, l) : -1;
}
int xvc_sched_poc_from_doc(int doc, int l) {
  return xvc_gpu::SubGop::Supported(l) ? xvc_gpu::SubGop::PocFromDoc(doc, l) : -1;
}
int xvc_sched_tid_from_doc(int doc, int l) {
  return xvc_gpu::SubGop::Supported(l) ? xvc_gpu::SubGop::TidFromDoc(doc, l) : -1;
}

int xvc_schedule_run(const xvc_schedule *s, int rank, const xvc_sched_callbacks *cb, void *user) {
  if (!s || !cb) return -1;
  return s->s.Run(rank, *cb, user);
}
int xvc_schedule_run_range(const xvc_schedule *s, int rank, const xvc_sched_callbacks *cb,
                           void *user, int first_op, int end_op) {
  if (!s || !cb) return -1;
  return s->s.Run(rank, *cb, user, first_op, end_op);
}

}  // extern "C"
