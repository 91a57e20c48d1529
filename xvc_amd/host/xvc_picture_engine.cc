// xvc_picture_engine.cc -- see xvc_picture_engine.h.
#include "xvc_picture_engine.h"

#include <new>
#include <vector>

namespace {
// Entry of picture `index` in a ring, safe to overwrite once `wait` has been applied to
// the event of whoever wrote what it held (a picture nobody referenced has no readers,
// and its slot may still be at work on the entry) and to every reader of it.
template <typename Wait>
int ClaimEntry(int ring, std::vector<int> *holds, std::vector<std::vector<xvcgpu_event *>> *readers,
               const std::vector<xvcgpu_event *> &ready, int index, Wait wait) {
  const int e = index % ring;
  if ((*holds)[e] >= 0) wait(ready[e]);
  for (xvcgpu_event *ev : (*readers)[e]) wait(ev);
  (*readers)[e].clear();
  (*holds)[e] = index;
  return e;
}
}  // namespace

struct xvc_picture_engine {
  xvc_picture_engine_desc d;
  const xvc_sched_picture *pictures;
  int n_pictures;
  std::vector<xvcgpu_event *> ready;                  // per ring entry
  std::vector<std::vector<xvcgpu_event *>> readers;   // events to wait for before overwriting
  std::vector<int> holds;
  std::vector<xvcgpu_event *> pool;                   // events of the sends
  size_t pool_next;
  int encoded;
  xvcgpu_status err;

  int IndexOfPoc(int poc) const {
    for (int i = 0; i < n_pictures; i++)
      if (pictures[i].poc == poc) return i;
    return -1;
  }
  // the pictures `p` lists, list 0 first, each once
  void Refs(const xvc_sched_picture &p, std::vector<int> *out) const {
    for (int l = 0; l < 2; l++)
      for (int k = 0; k < p.num_ref[l]; k++) {
        const int j = IndexOfPoc(p.ref_poc[l][k]);
        bool seen = false;
        for (int v : *out) seen |= v == j;
        if (!seen) out->push_back(j);
      }
  }
  template <typename Wait>
  int Claim(int index, Wait wait) {
    return ClaimEntry(d.ring, &holds, &readers, ready, index, wait);
  }
  bool Ok(xvcgpu_status s) {
    if (s != XVCGPU_OK && err == XVCGPU_OK) err = s;
    return s == XVCGPU_OK;
  }

  int Encode(const xvc_sched_picture &p, int index) {
    if (p.slot < 0 || p.slot >= d.n_slots) return XVCGPU_INVALID_ARGUMENT;
    xvcgpu_ctx *c = d.ctxs[p.slot];
    std::vector<int> refs;
    Refs(p, &refs);
    const int e = Claim(index, [&](xvcgpu_event *ev) { Ok(xvcgpu_event_wait(c, ev)); });
    if (p.intra) {
      // stands in for the intra picture of the segment: the padded original
      Ok(xvcgpu_picture_copy(c, d.recs[e], d.orig_of_picture[index]));
    } else {
      for (int j : refs) {
        if (j < 0 || holds[j % d.ring] != j) return XVCGPU_INVALID_ARGUMENT;
        Ok(xvcgpu_event_wait(c, ready[j % d.ring]));
      }
      if (refs.empty()) return XVCGPU_INVALID_ARGUMENT;
      xvcgpu_frame_pass_args a = *d.slot_args[p.slot];
      a.orig = d.orig_of_picture[index];
      a.ref = d.recs[refs[0] % d.ring];   // nearest list-0 picture: the frame pass's reference
      a.rec = d.recs[e];
      a.ref_poc = pictures[refs[0]].poc;
      Ok(xvcgpu_frame_pass(c, &a, XVC_FP_ENCODE | XVC_FP_DEBLOCK_V | XVC_FP_DEBLOCK_H |
                                      XVC_FP_PAD | XVC_FP_SSD));
    }
    Ok(xvcgpu_event_record(c, ready[e]));
    for (int j : refs) readers[j % d.ring].push_back(ready[e]);
    encoded++;
    if (err == XVCGPU_OK && d.after_encode) return d.after_encode(d.user, index);
    return err;
  }
  int Send(int index, int dst) {
    const int e = index % d.ring;
    if (!d.comm && d.host_send && holds[e] == index) {
      Ok(xvcgpu_event_wait(d.ctxs[0], ready[e]));
      if (err == XVCGPU_OK && d.host_send(d.user, e, dst)) return XVCGPU_DEVICE_ERROR;
      xvcgpu_event *ev = pool[pool_next++ % pool.size()];
      Ok(xvcgpu_event_record(d.ctxs[0], ev));
      readers[e].push_back(ev);
      return err;
    }
    if (!d.comm || holds[e] != index) return XVCGPU_INVALID_ARGUMENT;
    Ok(xvcgpu_comm_wait_event(d.comm, ready[e]));
    Ok(xvcgpu_comm_send_picture(d.comm, d.recs[e], dst));
    xvcgpu_event *ev = pool[pool_next++ % pool.size()];
    Ok(xvcgpu_comm_record_event(d.comm, ev));
    readers[e].push_back(ev);
    return err;
  }
  int Recv(int index, int src) {
    if (!d.comm && d.host_recv) {
      const int e = Claim(index, [&](xvcgpu_event *ev) { Ok(xvcgpu_event_wait(d.ctxs[0], ev)); });
      if (err == XVCGPU_OK && d.host_recv(d.user, e, src)) return XVCGPU_DEVICE_ERROR;
      Ok(xvcgpu_event_record(d.ctxs[0], ready[e]));
      return err;
    }
    if (!d.comm) return XVCGPU_INVALID_ARGUMENT;
    const int e = Claim(index, [&](xvcgpu_event *ev) { Ok(xvcgpu_comm_wait_event(d.comm, ev)); });
    Ok(xvcgpu_comm_recv_picture(d.comm, d.recs[e], src));
    Ok(xvcgpu_comm_record_event(d.comm, ready[e]));
    return err;
  }
};

namespace {
int CbEncode(void *user, const xvc_sched_picture *p, int index) {
  return static_cast<xvc_picture_engine *>(user)->Encode(*p, index);
}
int CbSend(void *user, const xvc_sched_picture *, int index, int dst) {
  return static_cast<xvc_picture_engine *>(user)->Send(index, dst);
}
int CbRecv(void *user, const xvc_sched_picture *, int index, int src) {
  return static_cast<xvc_picture_engine *>(user)->Recv(index, src);
}
}  // namespace

extern "C" {

xvc_picture_engine *xvc_host_picture_engine_create(const xvc_picture_engine_desc *desc) {
  if (!desc || !desc->schedule || desc->n_slots < 1 || !desc->ctxs || !desc->slot_args ||
      !desc->orig_of_picture || desc->ring < 1 || !desc->recs)
    return nullptr;
  xvc_picture_engine *e = new (std::nothrow) xvc_picture_engine();
  if (!e) return nullptr;
  e->d = *desc;
  e->pictures = xvc_schedule_pictures(desc->schedule);
  e->n_pictures = xvc_schedule_num_pictures(desc->schedule);
  e->pool_next = 0;
  e->encoded = 0;
  e->err = XVCGPU_OK;
  e->holds.assign(desc->ring, -1);
  e->readers.resize(desc->ring);
  bool ok = true;
  for (int i = 0; i < desc->ring && ok; i++) {
    xvcgpu_event *ev = nullptr;
    ok = xvcgpu_event_create(desc->ctxs[0], &ev) == XVCGPU_OK;
    e->ready.push_back(ev);
  }
  for (int i = 0; i < 4 * desc->ring && ok; i++) {
    xvcgpu_event *ev = nullptr;
    ok = xvcgpu_event_create(desc->ctxs[0], &ev) == XVCGPU_OK;
    e->pool.push_back(ev);
  }
  if (!ok) {
    xvc_host_picture_engine_destroy(e);
    return nullptr;
  }
  return e;
}

void xvc_host_picture_engine_destroy(xvc_picture_engine *e) {
  if (!e) return;
  for (xvcgpu_event *ev : e->ready)
    if (ev) xvcgpu_event_destroy(ev);
  for (xvcgpu_event *ev : e->pool)
    if (ev) xvcgpu_event_destroy(ev);
  delete e;
}

int xvc_host_picture_engine_run(xvc_picture_engine *e, int first_op, int end_op) {
  if (!e) return XVCGPU_INVALID_ARGUMENT;
  const xvc_sched_callbacks cb = {CbEncode, CbSend, CbRecv};
  return xvc_schedule_run_range(e->d.schedule, e->d.rank, &cb, e, first_op, end_op);
}

// The ring's ordering rule alone (no device): `n` claims of the pictures indices[i] on
// a ring whose entry e was written under event id 100 + e; before claim i the ids in
// reader_ids[i * 4 ..] (0-terminated, at most 4) are added as readers of that entry.
// waited_out receives, per claim, the ids waited for (0-terminated, 8 per claim).
int xvc_host_picture_ring_claims(int ring, int n, const int32_t *indices, const int32_t *reader_ids,
                                 int32_t *entries_out, int32_t *waited_out) {
  if (ring < 1 || n < 0 || !indices || !reader_ids || !entries_out || !waited_out) return -1;
  std::vector<int> holds(ring, -1);
  std::vector<std::vector<xvcgpu_event *>> readers(ring);
  std::vector<xvcgpu_event *> ready;
  for (int e = 0; e < ring; e++) ready.push_back(reinterpret_cast<xvcgpu_event *>(static_cast<intptr_t>(100 + e)));
  for (int i = 0; i < n; i++) {
    const int e = indices[i] % ring;
    for (int k = 0; k < 4 && reader_ids[4 * i + k]; k++)
      readers[e].push_back(reinterpret_cast<xvcgpu_event *>(static_cast<intptr_t>(reader_ids[4 * i + k])));
    int w = 0;
    for (int k = 0; k < 8; k++) waited_out[8 * i + k] = 0;
    entries_out[i] = ClaimEntry(ring, &holds, &readers, ready, indices[i], [&](xvcgpu_event *ev) {
      if (w < 8) waited_out[8 * i + w++] = static_cast<int32_t>(reinterpret_cast<intptr_t>(ev));
    });
    if (!readers[e].empty()) return -2;
  }
  return 0;
}

int xvc_host_picture_engine_holds(const xvc_picture_engine *e, int entry) {
  return e && entry >= 0 && entry < e->d.ring ? e->holds[entry] : -1;
}
int xvc_host_picture_engine_encoded(const xvc_picture_engine *e) { return e ? e->encoded : 0; }

}  // extern "C"
