/*
 * ref_stream.cc -- whole-stream side of the reference harness.
 *
 * TEST INFRASTRUCTURE ONLY (see ref_harness.cc): builds only where
 * /root/reference exists, into oracle/_ref/libxvcref.so.  No reference source
 * is kept here: the reference's encoder and decoder are driven through their
 * own public entry points, and two of the decoder's translation units
 * (xvc_dec_lib/cu_decoder.cc, picture_decoder.cc) are compiled as part of this
 * file, from where they lie, with two observation hooks:
 *   - every leaf CU is reported at the moment CuDecoder::DecompressCu marks
 *     it in the CU map (so the neighbour availability it sees is captured),
 *   - after every CTU the leaf CUs' final syntax (modes, vectors after
 *     CalculateMV, transform types, quantised levels) is copied out,
 *   - before / after in-loop filtering the reconstruction planes are copied.
 * tools/gen_stream_golden.py turns that into tests/golden/stream_*.npz: the
 * "parsed syntax" a decoder's reconstruction stage starts from, plus what it
 * must produce.
 */
#include <algorithm>
#include <array>
#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "xvc_enc_lib/xvcenc.h"
#include "xvc_dec_lib/xvcdec.h"

#define private public
#define protected public
#include "xvc_common_lib/checksum.h"
#include "xvc_common_lib/coding_unit.h"
#include "xvc_common_lib/picture_data.h"
#include "xvc_common_lib/quantize.h"
#include "xvc_common_lib/yuv_pic.h"
#include "xvc_dec_lib/cu_decoder.h"
#include "xvc_dec_lib/decoder.h"
#include "xvc_dec_lib/picture_decoder.h"
#undef private
#undef protected

namespace xr_stream {

/* One leaf CU as the reconstruction stage needs it.  Mirrored by the numpy
 * dtype STREAM_CU_DTYPE in tools/gen_stream_golden.py (size checked). */
struct Cu {
  int16_t x, y;           /* luma position */
  uint8_t w, h;           /* luma size */
  uint8_t tree;           /* 0 primary, 1 secondary (chroma tree, intra pictures) */
  uint8_t pred_mode;      /* 0 intra, 1 inter */
  int8_t qp[3];           /* CodingUnit::GetQp(comp): raw qp per component */
  uint8_t root_cbf;
  uint8_t cbf[3];
  uint8_t tx_skip[3];
  uint8_t dc_only[3];
  int8_t tx_select_idx;
  uint8_t tx_type[3][2];  /* GetTransformType(comp, 0 = vertical / 1 = horizontal) */
  int8_t intra_mode[3];   /* GetIntraMode(comp): 0..66, -2 = LM chroma */
  int8_t intra_chroma_raw;
  uint8_t inter_dir;      /* 0 L0, 1 L1, 2 bi */
  uint8_t skip, merge, affine, lic, fullpel;
  int8_t ref_idx[2];
  /* what IntraPrediction::DetermineNeighbors / the LIC model saw when this CU
   * was decompressed: per component XVC_INTRA_HAS_* bits, above-right and
   * below-left extents (in samples of the component) */
  uint8_t nb_flags[3];
  uint8_t nb_above_right[3];
  uint8_t nb_below_left[3];
  uint8_t depth;
  int32_t ref_poc[2];     /* -1: list unused */
  int32_t mv[2][4][2];    /* [list][corner][x, y], 1/16 pel, after CalculateMV */
  uint32_t level_off[3];  /* offset (in int16) of the comp's w*h levels, if cbf */
};

struct Info {
  int32_t poc, doc, tid, nal_type, pic_type; /* pic_type 0 intra 1 uni 2 bi */
  int32_t pic_qp, deblock, beta_offset, tc_offset, allow_lic, adaptive_qp;
  int32_t highest_layer, padded, width, height, bitdepth, two_trees;
  int32_t num_ref[2];
  int32_t ref_poc[2][5];
  int32_t n_cus;
  int32_t n_levels;
  uint8_t md5[16];
  int32_t conforming;
};

struct Picture {
  Info info;
  std::vector<Cu> cus;
  std::vector<int16_t> levels;
  std::vector<uint16_t> pre[3];   /* before deblocking, visible area */
  std::vector<uint16_t> post[3];  /* final, visible area */
};

static std::vector<Picture> g_pics;
static Picture *g_cur = nullptr;
static std::vector<const xvc::CodingUnit *> g_pending;  /* leaves of this CTU */
static int g_keep_planes = 1;

static void CopyPlanes(const xvc::YuvPicture &pic, std::vector<uint16_t> out[3]) {
  for (int c = 0; c < 3; c++) {
    xvc::YuvComponent comp = xvc::YuvComponent(c);
    const int w = pic.GetWidth(comp), h = pic.GetHeight(comp);
    out[c].resize(static_cast<size_t>(w) * h);
    for (int y = 0; y < h; y++) {
      std::memcpy(&out[c][static_cast<size_t>(y) * w], pic.GetSamplePtr(comp, 0, y),
                  sizeof(uint16_t) * w);
    }
  }
}

/* hook A: called right after PictureData::MarkUsedInPic(cu) in DecompressCu */
static void LeafMarked(xvc::CodingUnit *cu) {
  Cu rec;
  std::memset(&rec, 0, sizeof(rec));
  const auto comps = cu->GetPicData()->GetComponents(cu->GetCuTree());
  for (xvc::YuvComponent comp : comps) {
    const int c = static_cast<int>(comp);
    const int x = cu->GetPosX(comp), y = cu->GetPosY(comp);
    /* IntraPrediction::DetermineNeighbors, from the same public accessors */
    int flags = 0;
    if (x > 0) {
      flags |= 4;
      rec.nb_below_left[c] = static_cast<uint8_t>(cu->GetCuSizeBelowLeft(comp));
    }
    if (y > 0) {
      flags |= 2;
      rec.nb_above_right[c] = static_cast<uint8_t>(cu->GetCuSizeAboveRight(comp));
    }
    if (x > 0 && y > 0) flags |= 1;
    rec.nb_flags[c] = static_cast<uint8_t>(flags);
  }
  g_cur->cus.push_back(rec);
  g_pending.push_back(cu);
}

/* hook B: after CuDecoder::DecodeCtu - the CTU's leaves now hold their final
 * state and the CTU coefficient buffer still holds their levels */
static void CtuDone() {
  const size_t first = g_cur->cus.size() - g_pending.size();
  for (size_t k = 0; k < g_pending.size(); k++) {
    const xvc::CodingUnit *cu = g_pending[k];
    Cu &r = g_cur->cus[first + k];
    const xvc::YuvComponent luma = xvc::YuvComponent::kY;
    r.x = static_cast<int16_t>(cu->GetPosX(luma));
    r.y = static_cast<int16_t>(cu->GetPosY(luma));
    r.w = static_cast<uint8_t>(cu->GetWidth(luma));
    r.h = static_cast<uint8_t>(cu->GetHeight(luma));
    r.tree = static_cast<uint8_t>(cu->GetCuTree());
    r.pred_mode = static_cast<uint8_t>(cu->GetPredMode());
    r.depth = static_cast<uint8_t>(cu->GetDepth());
    r.root_cbf = cu->GetRootCbf();
    r.tx_select_idx = static_cast<int8_t>(cu->GetTransformSelectIdx());
    r.intra_mode[0] = r.intra_mode[1] = r.intra_mode[2] = -1;
    r.intra_chroma_raw = static_cast<int8_t>(cu->GetIntraChromaMode());
    const auto comps = cu->GetPicData()->GetComponents(cu->GetCuTree());
    for (xvc::YuvComponent comp : comps) {
      const int c = static_cast<int>(comp);
      r.qp[c] = static_cast<int8_t>(cu->GetQp(comp));
      r.cbf[c] = cu->GetCbf(comp);
      r.tx_skip[c] = cu->GetTransformSkip(comp);
      r.dc_only[c] = cu->GetDcCoeffOnly(comp);
      r.tx_type[c][0] = static_cast<uint8_t>(cu->GetTransformType(comp, 0));
      r.tx_type[c][1] = static_cast<uint8_t>(cu->GetTransformType(comp, 1));
      if (cu->IsIntra()) r.intra_mode[c] = static_cast<int8_t>(cu->GetIntraMode(comp));
      if (r.cbf[c]) {
        const int w = cu->GetWidth(comp), h = cu->GetHeight(comp);
        r.level_off[c] = static_cast<uint32_t>(g_cur->levels.size());
        xvc::DataBuffer<const xvc::Coeff> buf = cu->GetCoeff(comp);
        for (int y = 0; y < h; y++) {
          const xvc::Coeff *row = buf.GetDataPtr() + y * buf.GetStride();
          g_cur->levels.insert(g_cur->levels.end(), row, row + w);
        }
      }
    }
    r.ref_poc[0] = r.ref_poc[1] = -1;
    r.ref_idx[0] = r.ref_idx[1] = -1;
    if (cu->IsInter()) {
      r.inter_dir = static_cast<uint8_t>(cu->GetInterDir());
      r.skip = cu->GetSkipFlag();
      r.merge = cu->GetMergeFlag();
      r.affine = cu->GetUseAffine();
      r.lic = cu->GetUseLic();
      r.fullpel = cu->GetFullpelMv();
      for (int l = 0; l < 2; l++) {
        xvc::RefPicList list = static_cast<xvc::RefPicList>(l);
        /* the vectors of an unused list are copied as well: the in-loop filter
         * compares them when both CUs leave the same list unused
         * (deblocking_filter.cc:185-222) */
        for (int k4 = 0; k4 < 4; k4++) {
          const xvc::MotionVector &mv = cu->GetMv(list, static_cast<xvc::MvCorner>(k4));
          r.mv[l][k4][0] = mv.x;
          r.mv[l][k4][1] = mv.y;
        }
        if (!cu->HasMv(list)) continue;
        r.ref_idx[l] = static_cast<int8_t>(cu->GetRefIdx(list));
        r.ref_poc[l] = static_cast<int32_t>(cu->GetRefPoc(list));
      }
    }
  }
  g_pending.clear();
}

}  // namespace xr_stream

/* ---- the decoder's CU-level translation unit, with hook A ------------------ */
#define MarkUsedInPic(cu) MarkUsedInPic(cu); xr_stream::LeafMarked(cu)
#include "xvc_dec_lib/cu_decoder.cc"
#undef MarkUsedInPic

/* ---- the picture-level translation unit, with the CuDecoder it creates
 *      replaced by an observing wrapper -------------------------------------- */
namespace xvc {
class ObservedCuDecoder {
public:
  ObservedCuDecoder(const SimdFunctions &simd, YuvPicture *decoded_pic, PictureData *pic_data)
      : inner_(simd, decoded_pic, pic_data), pic_(decoded_pic), data_(pic_data) {
    xr_stream::g_pics.emplace_back();
    xr_stream::g_cur = &xr_stream::g_pics.back();
    xr_stream::Info &i = xr_stream::g_cur->info;
    std::memset(&i, 0, sizeof(i));
    i.poc = static_cast<int32_t>(data_->GetPoc());
    i.doc = static_cast<int32_t>(data_->GetDoc());
    i.tid = data_->GetTid();
    i.nal_type = static_cast<int32_t>(data_->GetNalType());
    i.pic_type = static_cast<int32_t>(data_->GetPredictionType());
    i.pic_qp = data_->GetPicQp()->GetQpRaw(YuvComponent::kY);
    i.deblock = data_->GetDeblock();
    i.beta_offset = data_->GetBetaOffset();
    i.tc_offset = data_->GetTcOffset();
    i.allow_lic = data_->GetUseLocalIlluminationCompensation();
    i.adaptive_qp = data_->GetAdaptiveQp();
    i.highest_layer = data_->IsHighestLayer();
    i.padded = data_->GetTid() == 0 || !data_->IsHighestLayer();
    i.width = pic_->GetWidth(YuvComponent::kY);
    i.height = pic_->GetHeight(YuvComponent::kY);
    i.bitdepth = pic_->GetBitdepth();
    i.two_trees = data_->HasSecondaryCuTree();
    const ReferencePictureLists *rpl = data_->GetRefPicLists();
    for (int l = 0; l < 2; l++) {
      RefPicList list = static_cast<RefPicList>(l);
      i.num_ref[l] = rpl->GetNumRefPics(list);
      for (int k = 0; k < 5; k++) {
        i.ref_poc[l][k] = k < i.num_ref[l] ? static_cast<int32_t>(rpl->GetRefPoc(list, k)) : -1;
      }
    }
  }
  ~ObservedCuDecoder() {
    /* end of PictureDecoder::Decode: filtered, padded, checksum validated */
    xr_stream::Picture *p = xr_stream::g_cur;
    if (xr_stream::g_keep_planes) xr_stream::CopyPlanes(*pic_, p->post);
    Checksum checksum(Checksum::kDefaultMethod, Checksum::Mode::kMinOverhead);
    checksum.HashPicture(*pic_);
    std::vector<uint8_t> hash = checksum.GetHash();
    std::memcpy(p->info.md5, hash.data(), 16);
    p->info.n_cus = static_cast<int32_t>(p->cus.size());
    p->info.n_levels = static_cast<int32_t>(p->levels.size());
  }
  void DecodeCtu(int rsaddr, SyntaxReader *reader) {
    inner_.DecodeCtu(rsaddr, reader);
    xr_stream::CtuDone();
    if (rsaddr == data_->GetNumberOfCtu() - 1 && xr_stream::g_keep_planes) {
      xr_stream::CopyPlanes(*pic_, xr_stream::g_cur->pre);
    }
  }

private:
  CuDecoder inner_;
  YuvPicture *pic_;
  PictureData *data_;
};
}  // namespace xvc

#define CuDecoder ObservedCuDecoder
#include "xvc_dec_lib/picture_decoder.cc"
#undef CuDecoder

extern "C" {

/* Encode n_frames packed planar 4:2:0 frames (8 bit: 1 byte per sample, else 2
 * LE) with the reference encoder through its public C API, xvcenc's defaults
 * plus the given qp / threads / (optional) explicit settings.  The NAL units
 * are appended to `out` in output order, each preceded by its 4-byte LE size
 * (the container xvcenc writes).  Returns the number of bytes, or -1. */
long xr_stream_encode(int width, int height, int input_bitdepth, int internal_bitdepth,
                      double framerate, int qp, int sub_gop_length, int speed_mode,
                      int tune_mode, int threads, const char *explicit_settings, int n_frames,
                      const uint8_t *frames, uint8_t *out, long out_cap) {
  const xvc_encoder_api *api = xvc_encoder_api_get();
  xvc_encoder_parameters *p = api->parameters_create();
  api->parameters_set_default(p);
  p->width = width;
  p->height = height;
  p->chroma_format = XVC_ENC_CHROMA_FORMAT_420;
  p->input_bitdepth = input_bitdepth;
  if (internal_bitdepth > 0) p->internal_bitdepth = internal_bitdepth;
  p->framerate = framerate;
  p->qp = qp;
  if (sub_gop_length > 0) p->sub_gop_length = sub_gop_length;
  if (speed_mode >= 0) p->speed_mode = speed_mode;
  if (tune_mode >= 0) p->tune_mode = tune_mode;
  p->threads = threads;
  std::string settings = explicit_settings ? explicit_settings : "";
  if (!settings.empty()) p->explicit_encoder_settings = &settings[0];
  if (api->parameters_check(p) != XVC_ENC_OK) {
    api->parameters_destroy(p);
    return -1;
  }
  xvc_encoder *enc = api->encoder_create(p);
  if (!enc) {
    api->parameters_destroy(p);
    return -1;
  }
  xvc_enc_pic_buffer *rec = api->picture_create(enc);
  const size_t frame_bytes =
      static_cast<size_t>(width) * height * 3 / 2 * (input_bitdepth > 8 ? 2 : 1);
  long used = 0;
  bool overflow = false;
  auto emit = [&](xvc_enc_nal_unit *nals, int n) {
    for (int i = 0; i < n; i++) {
      const uint32_t sz = static_cast<uint32_t>(nals[i].size);
      if (used + 4 + static_cast<long>(sz) > out_cap) {
        overflow = true;
        return;
      }
      out[used + 0] = sz & 0xff;
      out[used + 1] = (sz >> 8) & 0xff;
      out[used + 2] = (sz >> 16) & 0xff;
      out[used + 3] = (sz >> 24) & 0xff;
      std::memcpy(out + used + 4, nals[i].bytes, sz);
      used += 4 + sz;
    }
  };
  xvc_enc_nal_unit *nals = nullptr;
  int n = 0;
  for (int f = 0; f < n_frames; f++) {
    api->encoder_encode(enc, frames + f * frame_bytes, &nals, &n, rec);
    emit(nals, n);
  }
  while (api->encoder_flush(enc, &nals, &n, rec) == XVC_ENC_OK) emit(nals, n);
  emit(nals, n);
  api->picture_destroy(rec);
  api->encoder_destroy(enc);
  api->parameters_destroy(p);
  return overflow ? -1 : used;
}

/* Decode a stream in that container with the reference decoder (one thread)
 * and keep, per picture in decoding order, the leaf CUs' syntax, levels and the
 * reconstruction before / after the in-loop filter.  Returns the number of
 * pictures (negative: the decoder reported a checksum mismatch). */
int xr_stream_decode(const uint8_t *stream, long size, int keep_planes) {
  xr_stream::g_pics.clear();
  xr_stream::g_cur = nullptr;
  xr_stream::g_keep_planes = keep_planes;
  xvc::Decoder dec(0);
  xvc_decoded_picture out;
  long pos = 0;
  while (pos + 4 <= size) {
    const uint32_t sz = stream[pos] | (stream[pos + 1] << 8) | (stream[pos + 2] << 16) |
                        (static_cast<uint32_t>(stream[pos + 3]) << 24);
    pos += 4;
    if (pos + static_cast<long>(sz) > size) break;
    dec.DecodeNal(stream + pos, sz);
    pos += sz;
    while (dec.GetDecodedPicture(&out)) {
    }
  }
  dec.FlushBufferedNalUnits();
  while (dec.GetDecodedPicture(&out)) {
  }
  const int n = static_cast<int>(xr_stream::g_pics.size());
  for (auto &p : xr_stream::g_pics) p.info.conforming = dec.GetNumCorruptedPics() == 0;
  return dec.GetNumCorruptedPics() == 0 ? n : -n;
}

int xr_stream_cu_size(void) { return static_cast<int>(sizeof(xr_stream::Cu)); }
int xr_stream_info_size(void) { return static_cast<int>(sizeof(xr_stream::Info)); }

void xr_stream_get_info(int pic, void *out) {
  std::memcpy(out, &xr_stream::g_pics[pic].info, sizeof(xr_stream::Info));
}
void xr_stream_get_cus(int pic, void *out) {
  const auto &v = xr_stream::g_pics[pic].cus;
  std::memcpy(out, v.data(), v.size() * sizeof(xr_stream::Cu));
}
void xr_stream_get_levels(int pic, int16_t *out) {
  const auto &v = xr_stream::g_pics[pic].levels;
  std::memcpy(out, v.data(), v.size() * sizeof(int16_t));
}
/* which: 0 = before deblocking, 1 = final */
void xr_stream_get_plane(int pic, int which, int comp, uint16_t *out) {
  const auto &v = which ? xr_stream::g_pics[pic].post[comp] : xr_stream::g_pics[pic].pre[comp];
  std::memcpy(out, v.data(), v.size() * sizeof(uint16_t));
}
void xr_stream_release(void) {
  xr_stream::g_pics.clear();
  xr_stream::g_pics.shrink_to_fit();
}

}  // extern "C"
