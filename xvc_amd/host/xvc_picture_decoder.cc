// xvc_picture_decoder.cc -- see xvc_picture_decoder.h.
#include "xvc_picture_decoder.h"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

namespace xvc_gpu {

static double NowSeconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}


namespace {

// The 4x4 cell table of PictureData (picture_data.cc:52-62: one extra column /
// row beyond the CTU-aligned picture), one per CU tree, holding coding-order
// indices.  Cells are marked as the walk over the CU list proceeds, so a lookup
// sees exactly the CUs decoded before the current one - PictureData::GetCuAt at
// the time the reference decompresses that CU.
struct CellMap {
  int stride, rows;
  std::vector<int32_t> *cell;
  int At(int tree, int x, int y) const {
    if (x < 0 || y < 0) return -1;
    const int cx = x >> 2, cy = y >> 2;
    if (cx >= stride || cy >= rows) return -1;
    return cell[tree][static_cast<size_t>(cy) * stride + cx];
  }
};

inline int Max(int a, int b) { return a > b ? a : b; }

inline bool IsBlockSize(int v) { return v == 4 || v == 8 || v == 16 || v == 32 || v == 64; }

}  // namespace

// The syntax comes from a bitstream: nothing in it may index outside the cell
// map, the reference table or the level array.  (The reference asserts the same
// invariants while parsing - cu_reader.cc; here they are one check per CU
// before anything is planned.)
bool PictureDecoder::Validate(const xvc_picture_syntax &ps, const xvc_cu_syntax *cus,
                              const int16_t *levels) {
  if (ps.width <= 0 || ps.height <= 0 || ps.width > 16384 || ps.height > 16384 || ps.n_cus <= 0 ||
      !cus || ps.n_levels < 0 || (ps.n_levels > 0 && !levels))
    return false;
  if (ps.pic_type != XVC_PIC_BI && ps.pic_type != XVC_PIC_UNI && ps.pic_type != XVC_PIC_INTRA)
    return false;
  for (int l = 0; l < 2; l++)
    if (ps.num_ref[l] < 0 || ps.num_ref[l] > 5) return false;
  // CTU-aligned picture (the cell map holds one more column / row than that)
  const int aw = (ps.width + 63) & ~63, ah = (ps.height + 63) & ~63;
  const bool intra_pic = ps.pic_type == XVC_PIC_INTRA;
  for (int i = 0; i < ps.n_cus; i++) {
    const xvc_cu_syntax &cu = cus[i];
    if (cu.tree > 1 || (cu.tree == 1 && !intra_pic)) return false;
    if (!IsBlockSize(cu.w) || !IsBlockSize(cu.h)) return false;
    if (cu.x < 0 || cu.y < 0 || (cu.x & 3) || (cu.y & 3) || cu.x + cu.w > aw || cu.y + cu.h > ah)
      return false;
    if (cu.pred_mode > 1) return false;
    const bool has_luma = cu.tree == 0, has_chroma = cu.tree == 1 || !intra_pic;
    if (cu.pred_mode == 1) {
      if (intra_pic || cu.inter_dir > 2) return false;
      for (int l = 0; l < 2; l++) {
        const bool used = cu.inter_dir == 2 || cu.inter_dir == l;
        if (used && (cu.ref_idx[l] < 0 || cu.ref_idx[l] >= ps.num_ref[l])) return false;
      }
      if ((cu.flags & XVC_CU_AFFINE) && (cu.flags & XVC_CU_LIC)) return false;
    } else {
      for (int c = has_luma ? 0 : 1; c < (has_chroma ? 3 : 1); c++)
        if (cu.intra_mode[c] != XVC_CU_INTRA_LM &&
            (cu.intra_mode[c] < 0 || cu.intra_mode[c] >= XVC_INTRA_NUM_MODES))
          return false;
      if (has_luma && cu.intra_mode[0] == XVC_CU_INTRA_LM) return false;
    }
    for (int c = has_luma ? 0 : 1; c < (has_chroma ? 3 : 1); c++) {
      if (cu.tx_type[c][0] > XVC_TX_DST7 || cu.tx_type[c][1] > XVC_TX_DST7) return false;
      if (!cu.cbf[c]) continue;
      const int cs = c ? 1 : 0;
      const uint64_t end = static_cast<uint64_t>(cu.level_off[c]) +
                           static_cast<uint64_t>(cu.w >> cs) * (cu.h >> cs);
      if (end > static_cast<uint64_t>(ps.n_levels)) return false;
    }
  }
  return true;
}

void PictureDecoder::Plan(const xvc_picture_syntax &ps, const xvc_cu_syntax *cus,
                          const int16_t *levels, PicturePlan *plan) {
  const int n = ps.n_cus;
  PicturePlan &p = *plan;
  p.map_stride = (ps.width + 63) / 4 + 1;
  p.map_rows = (ps.height + 63) / 4 + 1;
  const size_t cells = static_cast<size_t>(p.map_stride) * p.map_rows;
  p.cell[0].assign(cells, -1);
  // the second tree's map only where a second tree exists (intra pictures): it is
  // half a megabyte to clear and to upload at 1080p
  bool second_tree = false;
  for (int i = 0; i < n && !second_tree; i++) second_tree = cus[i].tree != 0;
  if (second_tree)
    p.cell[1].assign(cells, -1);
  else
    p.cell[1].clear();
  p.wave.assign(n, 0);
  p.neighbors.assign(n, CuNeighbors());
  p.cu_info.assign(n, xvcgpu_cu_info());
  p.two_trees = false;
  CellMap map = {p.map_stride, p.map_rows, p.cell};
  // wave in which the CU's luma / chroma reconstruction is complete
  std::vector<int32_t> wl(n, 0), wc(n, 0);
  const bool intra_pic = ps.pic_type == XVC_PIC_INTRA;

  // reference table: one slot per distinct (list, idx); the caller's pictures
  // are matched to slots in Decode
  p.n_ref_slots = 0;
  for (int l = 0; l < 2; l++)
    for (int k = 0; k < 5; k++) p.ref_slot[l][k] = k < ps.num_ref[l] ? p.n_ref_slots++ : -1;

  struct Unit {  // one schedulable piece: the luma or the chroma of a CU
    int cu, wave, c0, c1;
  };
  std::vector<Unit> units;
  units.reserve(2 * n);

  for (int i = 0; i < n; i++) {
    const xvc_cu_syntax &cu = cus[i];
    const int tree = cu.tree;
    if (tree) p.two_trees = true;
    // PictureData::MarkUsedInPic (picture_data.cc:191-210), before the lookups
    for (int yy = 0; yy < cu.h; yy += 4) {
      int32_t *row = &p.cell[tree][static_cast<size_t>((cu.y + yy) >> 2) * p.map_stride +
                                   (cu.x >> 2)];
      std::fill(row, row + (cu.w >> 2), i);
    }
    const bool has_luma = tree == 0;
    const bool has_chroma = tree == 1 || !intra_pic;
    const bool intra = cu.pred_mode == 0;
    int dep_l = -1, dep_c = -1;  // latest wave this CU's luma / chroma waits for
    auto need = [&](int j) {
      if (j < 0) return;
      if (has_luma) dep_l = Max(dep_l, wl[j]);
      if (has_chroma) dep_c = Max(dep_c, wc[j]);
    };
    if (intra) {
      // IntraPrediction::DetermineNeighbors (intra_prediction.cc:688-705) with
      // GetCuSizeAboveRight / GetCuSizeBelowLeft (coding_unit.cc:304-336)
      int ar = 0, bl = 0;  // in luma samples
      if (cu.x > 0)
        for (int k = cu.w; k >= 0; k -= 4)
          if (map.At(tree, cu.x - 4, cu.y + cu.h - 4 + k) >= 0) {
            bl = k;
            break;
          }
      if (cu.y > 0)
        for (int k = cu.h; k >= 0; k -= 4)
          if (map.At(tree, cu.x + cu.w - 4 + k, cu.y - 4) >= 0) {
            ar = k;
            break;
          }
      CuNeighbors &nb = p.neighbors[i];
      for (int c = 0; c < 3; c++) {
        const int cs = c ? 1 : 0;
        int f = 0;
        if ((cu.x >> cs) > 0) f |= XVC_INTRA_HAS_LEFT;
        if ((cu.y >> cs) > 0) f |= XVC_INTRA_HAS_ABOVE;
        if ((cu.x >> cs) > 0 && (cu.y >> cs) > 0) f |= XVC_INTRA_HAS_ABOVE_LEFT;
        nb.flags[c] = static_cast<uint8_t>(f);
        nb.above_right[c] = static_cast<uint8_t>(ar >> cs);
        nb.below_left[c] = static_cast<uint8_t>(bl >> cs);
      }
      // the CUs whose reconstruction the reference samples come from
      if (cu.x > 0 && cu.y > 0) need(map.At(tree, cu.x - 4, cu.y - 4));
      if (cu.y > 0)
        for (int xx = 0; xx < cu.w + ar; xx += 4) need(map.At(tree, cu.x + xx, cu.y - 4));
      if (cu.x > 0)
        for (int yy = 0; yy < cu.h + bl; yy += 4) need(map.At(tree, cu.x - 4, cu.y + yy));
      // LM chroma (PredLmChroma, intra_prediction.cc:560-686): the co-located
      // luma with its row above / column to the left, from the luma tree
      if (has_chroma &&
          (cu.intra_mode[1] == XVC_CU_INTRA_LM || cu.intra_mode[2] == XVC_CU_INTRA_LM)) {
        for (int yy = -4; yy < cu.h; yy += 4)
          for (int xx = -4; xx < cu.w; xx += 4) {
            const int j = map.At(0, cu.x + xx, cu.y + yy);
            if (j >= 0 && j != i) dep_c = Max(dep_c, wl[j]);
          }
      }
    } else if (cu.flags & XVC_CU_LIC) {
      // DeriveLicParams reads the row above and the column left of the block
      if (cu.y > 0)
        for (int xx = 0; xx < cu.w; xx += 4) need(map.At(tree, cu.x + xx, cu.y - 4));
      if (cu.x > 0)
        for (int yy = 0; yy < cu.h; yy += 4) need(map.At(tree, cu.x - 4, cu.y + yy));
    }
    if (has_luma) wl[i] = dep_l + 1;
    if (has_chroma) {
      wc[i] = dep_c + 1;
      // one tree: the CU's own luma comes first (DecompressCu's component order)
      if (has_luma && intra &&
          (cu.intra_mode[1] == XVC_CU_INTRA_LM || cu.intra_mode[2] == XVC_CU_INTRA_LM))
        wc[i] = Max(wc[i], wl[i] + 1);
    }
    p.wave[i] = Max(has_luma ? wl[i] : 0, has_chroma ? wc[i] : 0);
    if (has_luma) units.push_back(Unit{i, wl[i], 0, 1});
    if (has_chroma) units.push_back(Unit{i, wc[i], 1, 3});

    // what the in-loop filter reads (deblocking_filter.cc:79-241)
    xvcgpu_cu_info &o = p.cu_info[i];
    o.x = static_cast<uint16_t>(cu.x);
    o.y = static_cast<uint16_t>(cu.y);
    o.w = cu.w;
    o.h = cu.h;
    o.intra = intra;
    o.cbf_luma = cu.cbf[0];
    o.qp_y = cu.qp[0];
    o.qp_c = cu.qp[1];
    o.ref_idx0 = cu.ref_idx[0];
    for (int l = 0; l < 2; l++) {
      const bool used = !intra && (cu.inter_dir == 2 || cu.inter_dir == l);
      o.ref_poc[l] = used ? ps.ref_poc[l][cu.ref_idx[l]] : -1;
      const bool affine = cu.flags & XVC_CU_AFFINE;
      for (int k = 0; k < 3; k++) {
        o.mv[l][k][0] = cu.mv[l][affine ? k : 0][0];
        o.mv[l][k][1] = cu.mv[l][affine ? k : 0][1];
      }
      // CodingUnit::SetMv(MotionVector3) (coding_unit.h:268-275)
      o.mv[l][3][0] = affine ? cu.mv[l][1][0] + cu.mv[l][2][0] - cu.mv[l][0][0] : cu.mv[l][0][0];
      o.mv[l][3][1] = affine ? cu.mv[l][1][1] + cu.mv[l][2][1] - cu.mv[l][0][1] : cu.mv[l][0][1];
    }
  }

  // job lists grouped by wave (stable: coding order inside a wave)
  int n_waves = 0;
  for (const Unit &u : units) n_waves = Max(n_waves, u.wave + 1);
  p.n_waves = n_waves;
  std::vector<int32_t> n_inter(n_waves + 1, 0), n_intra(n_waves + 1, 0), n_tx(n_waves + 1, 0);
  for (const Unit &u : units) {
    const int k = u.c1 - u.c0;
    (cus[u.cu].pred_mode == 0 ? n_intra : n_inter)[u.wave + 1] += k;
    n_tx[u.wave + 1] += k;
  }
  for (int w = 0; w < n_waves; w++) {
    n_inter[w + 1] += n_inter[w];
    n_intra[w + 1] += n_intra[w];
    n_tx[w + 1] += n_tx[w];
  }
  p.inter_first = n_inter;
  p.intra_first = n_intra;
  p.tx_first = n_tx;
  p.inter.assign(n_inter[n_waves], xvcgpu_inter_block());
  p.intra.assign(n_intra[n_waves], xvcgpu_intra_block());
  p.tx.assign(n_tx[n_waves], xvcgpu_tx_block());
  p.tx_level_off.assign(n_tx[n_waves], 0);
  p.tx_nnz.assign(n_tx[n_waves], 0);
  std::vector<int32_t> at_inter(n_inter.begin(), n_inter.end() - 1),
      at_intra(n_intra.begin(), n_intra.end() - 1), at_tx(n_tx.begin(), n_tx.end() - 1);
  // the final map answers "which CU is above / left" for the LIC model
  for (const Unit &u : units) {
    const xvc_cu_syntax &cu = cus[u.cu];
    const bool intra = cu.pred_mode == 0;
    for (int c = u.c0; c < u.c1; c++) {
      const int cs = c ? 1 : 0;
      const int x = cu.x >> cs, y = cu.y >> cs, w = cu.w >> cs, h = cu.h >> cs;
      if (intra) {
        xvcgpu_intra_block &j = p.intra[at_intra[u.wave]++];
        j.x = static_cast<int16_t>(x);
        j.y = static_cast<int16_t>(y);
        j.w = static_cast<uint8_t>(w);
        j.h = static_cast<uint8_t>(h);
        j.comp = static_cast<uint8_t>(c);
        j.mode = cu.intra_mode[c] == XVC_CU_INTRA_LM ? XVC_INTRA_MODE_LM_CHROMA
                                                     : static_cast<uint8_t>(cu.intra_mode[c]);
        j.neighbors = p.neighbors[u.cu].flags[c];
        j.above_right = p.neighbors[u.cu].above_right[c];
        j.below_left = p.neighbors[u.cu].below_left[c];
        j.reserved = 0;
      } else {
        xvcgpu_inter_block &j = p.inter[at_inter[u.wave]++];
        std::memset(&j, 0, sizeof(j));
        j.x = cu.x;
        j.y = cu.y;
        j.w = cu.w;
        j.h = cu.h;
        j.comp = static_cast<uint8_t>(c);
        j.flags = static_cast<uint8_t>(((cu.flags & XVC_CU_AFFINE) ? XVC_INTER_AFFINE : 0) |
                                       ((cu.flags & XVC_CU_LIC) ? XVC_INTER_LIC : 0));
        for (int l = 0; l < 2; l++) {
          const bool used = cu.inter_dir == 2 || cu.inter_dir == l;
          j.ref[l] = static_cast<int8_t>(used ? p.ref_slot[l][cu.ref_idx[l]] : -1);
          std::memcpy(j.mv[l], cu.mv[l], sizeof(j.mv[l]));
        }
        if (cu.flags & XVC_CU_LIC) {
          // GetCodingUnitAbove / Left (coding_unit.cc:227-234, :275-282): always
          // earlier in coding order, so the final map gives the same answer
          const int ia = cu.y > 0 ? map.At(cu.tree, cu.x, cu.y - 4) : -1;
          const int il = cu.x > 0 ? map.At(cu.tree, cu.x - 4, cu.y) : -1;
          if (ia >= 0) {
            j.neighbors |= XVC_LIC_HAS_ABOVE;
            j.above_x = cus[ia].x;
            j.above_y = cus[ia].y;
          }
          if (il >= 0) {
            j.neighbors |= XVC_LIC_HAS_LEFT;
            j.left_x = cus[il].x;
            j.left_y = cus[il].y;
          }
        }
      }
      const int t = at_tx[u.wave]++;
      xvcgpu_tx_block &b = p.tx[t];
      b.x = static_cast<int16_t>(x);
      b.y = static_cast<int16_t>(y);
      b.w = static_cast<uint8_t>(w);
      b.h = static_cast<uint8_t>(h);
      b.comp = static_cast<uint8_t>(c);
      b.tx_hor = cu.tx_skip[c] ? static_cast<uint8_t>(XVC_TX_SKIP) : cu.tx_type[c][1];
      b.tx_ver = cu.tx_type[c][0];
      // can_dst_4x4 (transform.cc:88-90)
      b.dst4x4 = c == 0 && intra && cu.tx_type[c][0] == XVC_TX_DEFAULT &&
                 cu.tx_type[c][1] == XVC_TX_DEFAULT;
      b.qp = cu.qp[c];
      b.intra_pic = 0;
      int nnz = 0;
      if (cu.cbf[c]) {
        const int16_t *lv = levels + cu.level_off[c];
        for (int k = 0; k < w * h; k++) nnz += lv[k] != 0;
        p.tx_level_off[t] = cu.level_off[c];
      }
      p.tx_nnz[t] = nnz;
    }
  }
}

PictureDecoder::PictureDecoder(xvcgpu_ctx *ctx, int width, int height, int bitdepth)
    : ctx_(ctx), width_(width), height_(height), bitdepth_(bitdepth), pred_(nullptr),
      last_waves_(0), last_launches_(0),
      use_waves_kernel_(true), tail_min_waves_(6), next_host_(0) {
  if (const char *e = std::getenv("XVC_DEC_TAIL_MIN_WAVES")) tail_min_waves_ = std::atoi(e);
  xvcgpu_picture_create(ctx_, width, height, bitdepth, &pred_);
  for (int k = 0; k < 2; k++) {
    d_staging_[k] = nullptr;
    staging_cap_[k] = 0;
    kernels_done_[k] = nullptr;
    staging_used_[k] = false;
  }
  next_staging_ = 0;
  for (HostSlot &h : host_) {
    h.mem = nullptr;
    h.cap = 0;
    h.copied = nullptr;
    h.in_flight = false;
  }
}

PictureDecoder::~PictureDecoder() {
  lanes_.clear();
  for (SeqSlot &q : seq_slots_) {
    if (q.in_flight) xvcgpu_event_synchronize(q.copied);
    if (q.copied) xvcgpu_event_destroy(q.copied);
    if (q.mem) xvcgpu_host_free(ctx_, q.mem);
  }
  for (xvcgpu_event *e : pic_done_)
    if (e) xvcgpu_event_destroy(e);
  for (HostSlot &h : host_) {
    if (h.in_flight) xvcgpu_event_synchronize(h.copied);
    if (h.copied) xvcgpu_event_destroy(h.copied);
    if (h.mem) xvcgpu_host_free(ctx_, h.mem);
  }
  xvcgpu_sync(ctx_);
  for (int k = 0; k < 2; k++) {
    if (d_staging_[k]) xvcgpu_free(ctx_, d_staging_[k]);
    if (kernels_done_[k]) xvcgpu_event_destroy(kernels_done_[k]);
  }
  if (pred_) xvcgpu_picture_destroy(pred_);
}

xvcgpu_status PictureDecoder::AddLane(xvcgpu_ctx *ctx) {
  if (!ctx || ctx == ctx_) return XVCGPU_INVALID_ARGUMENT;
  for (const auto &l : lanes_)
    if (l->ctx_ == ctx) return XVCGPU_INVALID_ARGUMENT;
  std::unique_ptr<PictureDecoder> d(new (std::nothrow) PictureDecoder(ctx, width_, height_, bitdepth_));
  if (!d || !d->pred_) return XVCGPU_OUT_OF_MEMORY;
  lanes_.push_back(std::move(d));
  return XVCGPU_OK;
}

xvcgpu_status PictureDecoder::EnsureStaging(int k, size_t bytes) {
  xvcgpu_status st = XVCGPU_OK;
  if (!kernels_done_[k]) {
    st = xvcgpu_event_create(ctx_, &kernels_done_[k]);
    if (st != XVCGPU_OK) return st;
  }
  if (bytes <= staging_cap_[k]) return XVCGPU_OK;
  if (d_staging_[k]) {
    // (kernels of the picture two back may still read it)
    if (staging_used_[k]) xvcgpu_event_synchronize(kernels_done_[k]);
    xvcgpu_free(ctx_, d_staging_[k]);
  }
  d_staging_[k] = nullptr;
  staging_cap_[k] = 0;
  staging_used_[k] = false;
  const size_t cap = bytes + bytes / 4;
  st = xvcgpu_malloc(ctx_, cap, &d_staging_[k]);
  if (st == XVCGPU_OK) staging_cap_[k] = cap;
  return st;
}

xvcgpu_status PictureDecoder::AcquireHostSlot(size_t bytes, HostSlot **out) {
  HostSlot &h = host_[next_host_];
  next_host_ ^= 1;
  xvcgpu_status st = XVCGPU_OK;
  if (h.in_flight) {   // the upload queued two pictures ago
    st = xvcgpu_event_synchronize(h.copied);
    if (st != XVCGPU_OK) return st;
    h.in_flight = false;
  }
  if (!h.copied) {
    st = xvcgpu_event_create(ctx_, &h.copied);
    if (st != XVCGPU_OK) return st;
  }
  if (bytes > h.cap) {
    if (h.mem) xvcgpu_host_free(ctx_, h.mem);
    h.mem = nullptr;
    h.cap = 0;
    const size_t cap = bytes + bytes / 4;
    st = xvcgpu_host_alloc(ctx_, cap, &h.mem);
    if (st != XVCGPU_OK) return st;
    h.cap = cap;
  }
  *out = &h;
  return XVCGPU_OK;
}

xvcgpu_status PictureDecoder::Decode(const xvc_picture_syntax &ps, const xvc_cu_syntax *cus,
                                     const int16_t *levels,
                                     const xvcgpu_picture *const ref_pics[2][5],
                                     xvcgpu_picture *rec) {
  if (!pred_ || !rec || !cus || ps.width != width_ || ps.height != height_ ||
      ps.bitdepth != bitdepth_ || ps.n_cus <= 0 || (ps.n_levels > 0 && !levels))
    return XVCGPU_INVALID_ARGUMENT;
  if (!Validate(ps, cus, levels)) return XVCGPU_INVALID_ARGUMENT;
  Plan(ps, cus, levels, &plan_);
  return Issue(plan_, ps, levels, ref_pics, rec);
}

// What a picture sends to the device in one piece - job lists, cell maps, CU records,
// levels, the wave table of its cooperative launch - and where each part lies in it.
void PictureDecoder::Layout(const PicturePlan &p, const xvc_picture_syntax &ps,
                            PackedPicture *L) const {
  // an intra picture (every wave holds intra jobs only, prediction job k and
  // transform job k are the same block): all waves in one cooperative launch
  L->one_launch = use_waves_kernel_ && p.inter.empty() && !p.intra.empty() &&
                  p.intra_first == p.tx_first;
  // an inter picture with intra CUs: from wave t0 on every job is an intra job (the
  // waves behind the inter CUs: intra CUs reading their neighbours' reconstruction) -
  // that tail goes into ONE cooperative launch too, instead of three launches per
  // wave (B pictures of the 1080p stream: 27-33 launches)
  int t0 = p.n_waves;
  while (t0 > 0 && p.inter_first[t0 - 1] == p.inter_first[p.n_waves]) t0--;
  L->t0 = t0;
  L->tail_launch = use_waves_kernel_ && !L->one_launch && t0 > 0 &&
                   p.n_waves - t0 >= tail_min_waves_;
  L->tail_first.clear();
  if (L->tail_launch)
    for (int w = t0; w <= p.n_waves; w++)
      L->tail_first.push_back(p.intra_first[w] - p.intra_first[t0]);
  const size_t bytes[10] = {
      p.inter.size() * sizeof(xvcgpu_inter_block),
      p.intra.size() * sizeof(xvcgpu_intra_block),
      p.tx.size() * sizeof(xvcgpu_tx_block),
      p.tx_level_off.size() * sizeof(uint32_t),
      p.tx_nnz.size() * sizeof(int32_t),
      p.cu_info.size() * sizeof(xvcgpu_cu_info),
      p.cell[0].size() * sizeof(int32_t),
      p.cell[1].size() * sizeof(int32_t),
      static_cast<size_t>(ps.n_levels > 0 ? ps.n_levels : 1) * sizeof(int16_t),
      (L->one_launch ? p.intra_first.size() : L->tail_first.size()) * sizeof(int32_t)};
  size_t total = 0;
  for (int k = 0; k < 10; k++) {
    L->bytes[k] = bytes[k];
    L->off[k] = total;
    total += (bytes[k] + 255) & ~static_cast<size_t>(255);
  }
  L->total = total;
}

void PictureDecoder::PackInto(const PicturePlan &p, const xvc_picture_syntax &ps,
                              const int16_t *levels, const PackedPicture &L, void *dst) {
  static const int16_t kNoLevels[1] = {0};
  const void *src[10] = {p.inter.data(), p.intra.data(), p.tx.data(), p.tx_level_off.data(),
                         p.tx_nnz.data(), p.cu_info.data(), p.cell[0].data(), p.cell[1].data(),
                         ps.n_levels > 0 ? static_cast<const void *>(levels) : kNoLevels,
                         L.one_launch ? p.intra_first.data() : L.tail_first.data()};
  for (int k = 0; k < 10; k++)
    if (L.bytes[k]) std::memcpy(static_cast<uint8_t *>(dst) + L.off[k], src[k], L.bytes[k]);
}

xvcgpu_status PictureDecoder::Issue(const PicturePlan &p, const xvc_picture_syntax &ps,
                                    const int16_t *levels,
                                    const xvcgpu_picture *const ref_pics[2][5],
                                    xvcgpu_picture *rec) {
  PackedPicture L;
  Layout(p, ps, &L);
  HostSlot *slot = nullptr;
  xvcgpu_status st = AcquireHostSlot(L.total, &slot);
  if (st != XVCGPU_OK) return st;
  PackInto(p, ps, levels, L, slot->mem);
  st = IssuePacked(p, ps, L, slot->mem, slot->copied, ref_pics, rec);
  slot->in_flight = true;
  return st;
}

// The picture's upload (from page-locked memory the caller filled: PackInto) and its
// launches.  `copied` is recorded behind the upload: the memory may be written again
// once it has passed.
xvcgpu_status PictureDecoder::IssuePacked(const PicturePlan &p, const xvc_picture_syntax &ps,
                                          const PackedPicture &L, const void *host_mem,
                                          xvcgpu_event *copied,
                                          const xvcgpu_picture *const ref_pics[2][5],
                                          xvcgpu_picture *rec) {
  const size_t *off = L.off;
  const bool one_launch = L.one_launch, tail_launch = L.tail_launch;
  const int t0 = L.t0;
  const size_t total = L.total;
  const int sk = next_staging_;
  next_staging_ ^= 1;
  xvcgpu_status st = EnsureStaging(sk, total);
  if (st != XVCGPU_OK) return st;
  // on the copy stream, behind the kernels of the picture two back (they read this
  // device buffer) - beside, not behind, the previous picture's kernels
  st = xvcgpu_upload_ahead(ctx_, d_staging_[sk], host_mem, total,
                           staging_used_[sk] ? kernels_done_[sk] : nullptr, copied);
  if (st != XVCGPU_OK) return st;
  st = xvcgpu_event_wait(ctx_, copied);   // this picture's kernels: after its upload
  if (st != XVCGPU_OK) return st;
  uint8_t *base = static_cast<uint8_t *>(d_staging_[sk]);
  const xvcgpu_inter_block *d_inter = reinterpret_cast<const xvcgpu_inter_block *>(base + off[0]);
  const xvcgpu_intra_block *d_intra = reinterpret_cast<const xvcgpu_intra_block *>(base + off[1]);
  const xvcgpu_tx_block *d_tx = reinterpret_cast<const xvcgpu_tx_block *>(base + off[2]);
  const uint32_t *d_off = reinterpret_cast<const uint32_t *>(base + off[3]);
  const int32_t *d_nnz = reinterpret_cast<const int32_t *>(base + off[4]);
  const xvcgpu_cu_info *d_cus = reinterpret_cast<const xvcgpu_cu_info *>(base + off[5]);
  const int32_t *d_cell0 = reinterpret_cast<const int32_t *>(base + off[6]);
  const int32_t *d_cell1 = reinterpret_cast<const int32_t *>(base + off[7]);
  const int16_t *d_levels = reinterpret_cast<const int16_t *>(base + off[8]);

  const xvcgpu_picture *refs[10];
  for (int l = 0; l < 2; l++)
    for (int k = 0; k < 5; k++)
      if (p.ref_slot[l][k] >= 0) {
        if (!ref_pics || !ref_pics[l][k]) return XVCGPU_INVALID_ARGUMENT;
        refs[p.ref_slot[l][k]] = ref_pics[l][k];
      }

  int launches = 0;
  bool waves_done = false;
  if (one_launch) {
    const int32_t *d_first = reinterpret_cast<const int32_t *>(base + off[9]);
    st = xvcgpu_intra_recon_waves(ctx_, rec, pred_, d_intra, d_tx, d_first, p.n_waves, d_levels,
                                  d_off, d_nnz);
    if (st == XVCGPU_OK) {
      waves_done = true;
      launches++;
    } else if (st != XVCGPU_UNSUPPORTED) {
      return st;
    }
  }
  int n_sep = p.n_waves;   // waves issued as separate launches
  if (tail_launch) n_sep = t0;
  for (int w = 0; w < n_sep && !waves_done; w++) {
    const int i0 = p.inter_first[w], i1 = p.inter_first[w + 1];
    if (i1 > i0) {
      st = xvcgpu_inter_pred_batch(ctx_, refs, p.n_ref_slots, rec, pred_, d_inter + i0, i1 - i0);
      if (st != XVCGPU_OK) return st;
      launches++;
    }
    const int a0 = p.intra_first[w], a1 = p.intra_first[w + 1];
    if (a1 > a0) {
      st = xvcgpu_intra_pred_batch(ctx_, rec, pred_, d_intra + a0, a1 - a0);
      if (st != XVCGPU_OK) return st;
      launches++;
    }
    const int t0 = p.tx_first[w], t1 = p.tx_first[w + 1];
    if (t1 > t0) {
      st = xvcgpu_inv_transform_batch(ctx_, pred_, rec, d_tx + t0, t1 - t0, d_levels, d_off + t0,
                                      d_nnz + t0);
      if (st != XVCGPU_OK) return st;
      launches += 2;
    }
  }
  if (tail_launch) {
    // (in these waves job k of the intra list and job tx_first[t0] + k of the
    // transform list are the same block: every unit is an intra unit)
    const int32_t *d_first = reinterpret_cast<const int32_t *>(base + off[9]);
    const int a = p.intra_first[t0], t = p.tx_first[t0];
    st = xvcgpu_intra_recon_waves(ctx_, rec, pred_, d_intra + a, d_tx + t, d_first,
                                  p.n_waves - t0, d_levels, d_off + t, d_nnz + t);
    if (st == XVCGPU_OK) {
      launches++;
    } else if (st == XVCGPU_UNSUPPORTED) {
      for (int w = t0; w < p.n_waves; w++) {
        const int a0 = p.intra_first[w], a1 = p.intra_first[w + 1];
        const int t0w = p.tx_first[w], t1w = p.tx_first[w + 1];
        st = xvcgpu_intra_pred_batch(ctx_, rec, pred_, d_intra + a0, a1 - a0);
        if (st != XVCGPU_OK) return st;
        st = xvcgpu_inv_transform_batch(ctx_, pred_, rec, d_tx + t0w, t1w - t0w, d_levels,
                                        d_off + t0w, d_nnz + t0w);
        if (st != XVCGPU_OK) return st;
        launches += 3;
      }
    } else {
      return st;
    }
  }
  if (ps.deblock) {
    const int bipic = ps.pic_type == XVC_PIC_BI;
    if (p.two_trees) {
      // deblocking_filter.cc:63-75: the primary tree filters luma on the
      // 4-sample grid, the secondary tree chroma on the 8-sample grid
      st = xvcgpu_deblock_tree(ctx_, rec, d_cus, ps.n_cus, d_cell0, p.map_stride, bipic,
                               ps.beta_offset, ps.tc_offset, 4, 1);
      if (st != XVCGPU_OK) return st;
      st = xvcgpu_deblock_tree(ctx_, rec, d_cus, ps.n_cus, d_cell1, p.map_stride, bipic,
                               ps.beta_offset, ps.tc_offset, 8, 2);
    } else {
      st = xvcgpu_deblock_tree(ctx_, rec, d_cus, ps.n_cus, d_cell0, p.map_stride, bipic,
                               ps.beta_offset, ps.tc_offset, 4, 3);
    }
    if (st != XVCGPU_OK) return st;
    launches += p.two_trees ? 4 : 2;
  }
  if (ps.pad_border) {
    st = xvcgpu_pad_border(ctx_, rec);
    if (st != XVCGPU_OK) return st;
    launches++;
  }
  st = xvcgpu_event_record(ctx_, kernels_done_[sk]);
  if (st != XVCGPU_OK) return st;
  staging_used_[sk] = true;
  last_waves_ = p.n_waves;
  last_launches_ = launches;
  return XVCGPU_OK;
}

// The pictures of a sequence, planning one ahead of issuing: while this thread uploads
// and launches picture i, a worker validates and plans picture i + 1 (pure host work,
// 0.1 ms per 1080p B picture - as long as the device takes for the picture).
xvcgpu_status PictureDecoder::DecodeSequence(int n, const xvc_picture_syntax *const *ps,
                                             const xvc_cu_syntax *const *cus,
                                             const int16_t *const *levels,
                                             const int32_t *ref_index,
                                             xvcgpu_picture *const *recs) {
  if (n < 0 || (n && (!ps || !cus || !levels || !ref_index || !recs)))
    return XVCGPU_INVALID_ARGUMENT;
  if (!n) return XVCGPU_OK;
  // With lanes a picture is ordered behind the pictures it REFERENCES only: a buffer
  // handed in twice (a recycled DPB entry) would be written on one lane while another
  // lane still reads what it held - refused (one lane keeps stream order and takes it)
  if (num_lanes() > 1)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++)
        if (recs[i] && recs[i] == recs[j]) return XVCGPU_INVALID_ARGUMENT;
  // kWorkers planners, each taking every kWorkers-th picture, kRing plan slots: a plan
  // costs about as much host time as a B picture takes on the device, and more when
  // its thread shares the memory system with the issuing thread - several in flight
  // keep the issuing thread from waiting
  enum { kWorkers = 4, kRing = 8 };
  if (seq_plans_.size() != kRing) seq_plans_.resize(kRing);
  std::vector<PicturePlan> &plans = seq_plans_;
  bool valid[kRing] = {};
  bool ready[kRing] = {};
  const bool trace_workers = std::getenv("XVC_DEC_TRACE") != nullptr;
  // XVC_DEC_PACK_ON_ISSUE=1: the issuing thread copies a picture's parts into its own two
  // page-locked slots (the form before the planners packed: for comparisons)
  const bool pack_on_issue = std::getenv("XVC_DEC_PACK_ON_ISSUE") != nullptr;
  std::mutex trace_mu;
  double w_plan = 0, w_pack = 0;
  if (seq_slots_.size() != kRing) {
    SeqSlot empty = {};
    empty.mem = nullptr;
    empty.cap = 0;
    empty.copied = nullptr;
    empty.in_flight = false;
    seq_slots_.assign(kRing, empty);
  }
  for (SeqSlot &q : seq_slots_) {
    if (!q.copied && xvcgpu_event_create(ctx_, &q.copied) != XVCGPU_OK) return XVCGPU_DEVICE_ERROR;
    if (q.in_flight) {          // an earlier sequence's upload
      xvcgpu_event_synchronize(q.copied);
      q.in_flight = false;
    }
  }
  // a planner validates, plans and PACKS its picture: the parts of the upload go into
  // the ring slot's page-locked memory on the planner's thread
  auto prepare = [&](int i) {
    const xvc_picture_syntax &s = *ps[i];
    const int k = i % kRing;
    valid[k] = cus[i] && s.width == width_ && s.height == height_ &&
               s.bitdepth == bitdepth_ && s.n_cus > 0 && (s.n_levels <= 0 || levels[i]) &&
               Validate(s, cus[i], levels[i]);
    if (!valid[k]) return;
    const double tp0 = trace_workers ? NowSeconds() : 0;
    Plan(s, cus[i], levels[i], &plans[k]);
    if (pack_on_issue) return;
    SeqSlot &q = seq_slots_[k];
    Layout(plans[k], s, &q.lay);
    // (the slot's previous upload has passed: the issuing thread said so - `uploaded` -
    // before this planner was let at the slot; planners make no runtime calls but the
    // occasional allocation)
    const double tp1 = trace_workers ? NowSeconds() : 0;
    if (q.lay.total > q.cap) {
      if (q.mem) xvcgpu_host_free(ctx_, q.mem);
      q.mem = nullptr;
      q.cap = 0;
      const size_t cap = q.lay.total + q.lay.total / 4;
      if (xvcgpu_host_alloc(ctx_, cap, &q.mem) != XVCGPU_OK) {
        valid[k] = false;
        return;
      }
      q.cap = cap;
    }
    PackInto(plans[k], s, levels[i], q.lay, q.mem);
    if (trace_workers) {
      const double tp3 = NowSeconds();
      std::lock_guard<std::mutex> lk(trace_mu);
      w_plan += tp1 - tp0;
      w_pack += tp3 - tp1;
    }
  };
  // slot i % kRing may be written once picture i - kRing (its previous user) has been
  // issued; picture i is issued once its plan is there
  std::mutex mu;
  std::condition_variable cv;
  int issued = 0;
  int uploaded = -1;     // the uploads of the pictures up to this one have passed
  bool stop = false;
  std::vector<std::thread> workers;
  for (int wk = 0; wk < kWorkers && wk < n; wk++)
    workers.emplace_back([&, wk]() {
      for (int i = wk; i < n; i += kWorkers) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&]() { return stop || (issued > i - kRing && uploaded >= i - kRing); });
          if (stop) return;
        }
        prepare(i);
        {
          std::lock_guard<std::mutex> lk(mu);
          ready[i % kRing] = true;
        }
        cv.notify_all();
      }
    });
  xvcgpu_status st = XVCGPU_OK;
  const int n_lanes = num_lanes();
  if (n_lanes > 1) {
    for (auto &l : lanes_) {
      l->use_waves_kernel_ = use_waves_kernel_;
      l->tail_min_waves_ = tail_min_waves_;
    }
    while (static_cast<int>(pic_done_.size()) < n && st == XVCGPU_OK) {
      xvcgpu_event *e = nullptr;
      st = xvcgpu_event_create(ctx_, &e);
      if (st == XVCGPU_OK) pic_done_.push_back(e);
    }
  }
  const bool trace = std::getenv("XVC_DEC_TRACE") != nullptr;
  double t_wait = 0, t_issue = 0;
  auto now = []() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  enum { kUploadsAhead = 4 };   // uploads in flight at most; the planners' slots behind them
  for (int i = 0; i < n && st == XVCGPU_OK; i++) {
    if (i >= kUploadsAhead) {
      // (the only wait for the device, and on this thread: usually long passed)
      SeqSlot &o = seq_slots_[(i - kUploadsAhead) % kRing];
      if (o.in_flight) {
        xvcgpu_event_synchronize(o.copied);
        o.in_flight = false;
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        uploaded = i - kUploadsAhead;
      }
      cv.notify_all();
    }
    const double ta = now();
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&]() { return ready[i % kRing]; });
      ready[i % kRing] = false;
    }
    const double tb = now();
    t_wait += tb - ta;
    if (!valid[i % kRing] || !recs[i]) {
      st = XVCGPU_INVALID_ARGUMENT;
    } else {
      const xvcgpu_picture *refs[2][5];
      for (int l = 0; l < 2 && st == XVCGPU_OK; l++)
        for (int k = 0; k < 5; k++) {
          const int j = ref_index[(i * 2 + l) * 5 + k];
          if (j >= i) st = XVCGPU_INVALID_ARGUMENT;   // only pictures decoded before this one
          refs[l][k] = j >= 0 && j < i ? recs[j] : nullptr;
        }
      // the picture's lane; its kernels behind the pictures it reads that ran elsewhere
      const int lane = i % n_lanes;
      PictureDecoder *pd = lane ? lanes_[lane - 1].get() : this;
      if (n_lanes > 1)
        for (int l = 0; l < 2 && st == XVCGPU_OK; l++)
          for (int k = 0; k < 5 && st == XVCGPU_OK; k++) {
            const int j = ref_index[(i * 2 + l) * 5 + k];
            if (j >= 0 && j % n_lanes != lane) st = xvcgpu_event_wait(pd->ctx_, pic_done_[j]);
          }
      if (st == XVCGPU_OK && pack_on_issue) {
        st = pd->Issue(plans[i % kRing], *ps[i], levels[i], refs, recs[i]);
      } else if (st == XVCGPU_OK) {
        SeqSlot &q = seq_slots_[i % kRing];
        st = pd->IssuePacked(plans[i % kRing], *ps[i], q.lay, q.mem, q.copied, refs, recs[i]);
        q.in_flight = true;
      }
      if (st == XVCGPU_OK && n_lanes > 1) st = xvcgpu_event_record(pd->ctx_, pic_done_[i]);
    }
    t_issue += now() - tb;
    {
      std::lock_guard<std::mutex> lk(mu);
      issued = i + 1;
      if (st != XVCGPU_OK) stop = true;
    }
    cv.notify_all();
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = stop || st != XVCGPU_OK;
  }
  cv.notify_all();
  for (std::thread &w : workers) w.join();
  // the caller waits on this decoder's context: it is behind every lane's last picture
  if (n_lanes > 1 && st == XVCGPU_OK)
    for (int lane = 1; lane < n_lanes && st == XVCGPU_OK; lane++) {
      const int last = ((n - 1 - lane) / n_lanes) * n_lanes + lane;
      if (last >= 0 && last < n && n - 1 >= lane) st = xvcgpu_event_wait(ctx_, pic_done_[last]);
    }
  else if (n_lanes > 1)
    for (auto &l : lanes_) xvcgpu_sync(l->ctx_);   // (an error: leave no lane running)
  if (trace)
    std::fprintf(stderr, "DecodeSequence: %d pictures, waiting for plans %.3f ms, issuing %.3f ms; "
                 "planners (summed): plan %.3f ms, packing %.3f ms\n",
                 n, 1e3 * t_wait, 1e3 * t_issue, 1e3 * w_plan, 1e3 * w_pack);
  return st;
}

}  // namespace xvc_gpu

struct xvc_host_picture_decoder {
  xvc_gpu::PictureDecoder dec;
  xvc_host_picture_decoder(xvcgpu_ctx *c, int w, int h, int bd) : dec(c, w, h, bd) {}
};

extern "C" {

xvc_host_picture_decoder *xvc_host_picture_decoder_create(xvcgpu_ctx *ctx, int width, int height,
                                                          int bitdepth) {
  if (!ctx) return nullptr;
  return new (std::nothrow) xvc_host_picture_decoder(ctx, width, height, bitdepth);
}

void xvc_host_picture_decoder_destroy(xvc_host_picture_decoder *d) { delete d; }

int xvc_host_picture_decoder_decode(xvc_host_picture_decoder *d, const xvc_picture_syntax *ps,
                                    const xvc_cu_syntax *cus, const int16_t *levels,
                                    const xvcgpu_picture *const *ref_pics, xvcgpu_picture *rec) {
  if (!d || !ps) return XVCGPU_INVALID_ARGUMENT;
  const xvcgpu_picture *refs[2][5];
  for (int l = 0; l < 2; l++)
    for (int k = 0; k < 5; k++) refs[l][k] = ref_pics ? ref_pics[l * 5 + k] : nullptr;
  return d->dec.Decode(*ps, cus, levels, refs, rec);
}

int xvc_host_picture_decoder_decode_sequence(xvc_host_picture_decoder *d, int n,
                                             const xvc_picture_syntax *const *ps,
                                             const xvc_cu_syntax *const *cus,
                                             const int16_t *const *levels,
                                             const int32_t *ref_index,
                                             xvcgpu_picture *const *recs) {
  if (!d) return XVCGPU_INVALID_ARGUMENT;
  return d->dec.DecodeSequence(n, ps, cus, levels, ref_index, recs);
}

void xvc_host_picture_decoder_one_launch_intra(xvc_host_picture_decoder *d, int on) {
  if (d) d->dec.set_one_launch_intra(on != 0);
}

int xvc_host_picture_decoder_add_lane(xvc_host_picture_decoder *d, xvcgpu_ctx *ctx) {
  return d ? d->dec.AddLane(ctx) : XVCGPU_INVALID_ARGUMENT;
}

int xvc_host_picture_decoder_waves(const xvc_host_picture_decoder *d) {
  return d ? d->dec.last_num_waves() : 0;
}
int xvc_host_picture_decoder_launches(const xvc_host_picture_decoder *d) {
  return d ? d->dec.last_num_launches() : 0;
}

int xvc_host_plan_picture(const xvc_picture_syntax *ps, const xvc_cu_syntax *cus,
                          const int16_t *levels, uint8_t *neighbors_out, int32_t *wave_out) {
  if (!ps || !cus) return -1;
  if (!xvc_gpu::PictureDecoder::Validate(*ps, cus, levels)) return -1;
  xvc_gpu::PicturePlan plan;
  xvc_gpu::PictureDecoder::Plan(*ps, cus, levels, &plan);
  for (int i = 0; i < ps->n_cus; i++) {
    if (neighbors_out) {
      for (int c = 0; c < 3; c++) {
        neighbors_out[9 * i + c] = plan.neighbors[i].flags[c];
        neighbors_out[9 * i + 3 + c] = plan.neighbors[i].above_right[c];
        neighbors_out[9 * i + 6 + c] = plan.neighbors[i].below_left[c];
      }
    }
    if (wave_out) wave_out[i] = plan.wave[i];
  }
  return plan.n_waves;
}

}  // extern "C"
