// xvc_cu_state_builder.cc -- see xvc_cu_state_builder.h.
#include "xvc_cu_state_builder.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <tuple>
#include <vector>

namespace xvc_gpu {
namespace {

const int R3 = XVC_CS_MAX_REFS;
const int kBiSlots = 2 * R3 * R3;
const int kSlot = 64;          // scratch geometry: slot k of a state at luma x = 64 k
const int kMaxSlots = 8;
const int kMergeSlots = XVC_CS_MERGE_SLOTS;
const int32_t kDevice = 0x7fffff;   // a field the device composes (the folds overwrite it)

// picture selectors of the ops (xvc_cu_state.h)
enum { PIC_ORIG = 0, PIC_S_ORIG, PIC_S_PRED, PIC_S_REC, PIC_NB, PIC_REC, PIC_IPRED, PIC_IREC };

struct StartGroup { int slot, first_dist, count, first_cand; };
struct AffStart { int first, n, start_dist; };
typedef std::pair<int, int> SlotJob;   // (picture slot, job)

template <typename T>
T Zero() {
  T v;
  std::memset(&v, 0, sizeof(v));
  return v;
}

}  // namespace

// The composer's state: outputs as vectors, the groupings the programs need.
class CuStateBuilder {
 public:
  int Build(const xvc_csb_picture &pic);
  const xvc_cs_op *Program(const xvc_csb_addrs &a, const xvc_csb_intra &in, int first, int n,
                           int flags, int64_t *n_ops);
  const void *Array(int which, int64_t *bytes) const;

  int n_start_dist = 0, n_bi_slots = 0;
  int64_t n_edist = 0;

 private:
  int BuildPasses();
  void BuildMergeFolds();
  void BuildEvalCands();

  xvc_csb_picture p_;
  std::vector<xvc_cs_state> states_;
  std::vector<uint8_t> call_comp_;
  std::vector<xvcgpu_cs_pass> passes_;
  std::vector<int64_t> pass_first_, pass_count_, aff_work_src_, merge_state_, edist_first_;
  std::vector<uint8_t> folded_, start_slots_, me_slots_, bi_slots_, aff_slots_;
  std::vector<xvcgpu_mc_metric_cand> start_cands_;
  std::vector<xvcgpu_inter_block> aff_start_inter_, ev_inter_work_, mg_slots_;
  std::vector<xvcgpu_block_pos> aff_start_dst_, call_pos_;
  std::vector<xvcgpu_metric_cand> aff_start_cands_;
  std::vector<xvcgpu_copy_block> aff_start_copy_;
  std::vector<xvcgpu_me_block> me_work_;
  std::vector<xvcgpu_mc_lic_block> bi_lic_work_;
  std::vector<xvcgpu_affine_me_block> aff_work_;
  std::vector<xvcgpu_cs_merge> mg_fold_;
  std::vector<xvcgpu_eval_cand> ev_cands_, ev_cands_copy_, mg_ecands_, aff_start_ecands_;
  std::map<int, std::vector<StartGroup> > start_groups_;
  std::map<int, std::vector<SlotJob> > uni_groups_, aff_uni_groups_;
  std::map<int, AffStart> aff_start_;
  std::vector<xvc_cs_op> ops_;
};

int CuStateBuilder::Build(const xvc_csb_picture &pic) {
  p_ = pic;
  states_.assign(pic.states, pic.states + pic.n_states);
  const int rc = BuildPasses();
  if (rc) return rc;
  BuildMergeFolds();
  BuildEvalCands();
  return 0;
}

// The passes of every folded inter / motion state and the arrays the chained form works on
// (SearchMotion inter_search.cc:199-259: per list and picture EvalStartMvp -> search ->
// EvalFinalMvpIdx -> cost fold, the refinement, the three-way choice; CompressInter :74-98
// runs it a second time with the affine model).
int CuStateBuilder::BuildPasses() {
  const int n_st = p_.n_states;
  int same[R3];
  for (int r = 0; r < R3; r++) {
    same[r] = -1;
    if (r < p_.n_ref[1])
      for (int q = 0; q < p_.n_ref[0]; q++)
        if (p_.ref_poc[0][q] == p_.ref_poc[1][r]) { same[r] = q; break; }
  }
  auto slot_of = [&](int poc) {
    for (int i = 0; i < p_.n_slots; i++)
      if (p_.slot_pocs[i] == poc) return i;
    return -1;
  };
  pass_first_.assign(n_st, -1);
  pass_count_.assign(n_st, 0);
  folded_.assign(n_st, 0);
  me_work_.assign(p_.me_jobs, p_.me_jobs + p_.n_me);
  std::vector<int64_t> aff_rows;          // source row in aff_jobs, -1: a refinement slot
  std::map<int, xvcgpu_mc_lic_block> bi_lic_rows;
  for (int mi = 0; mi < p_.n_motions; mi++) {
    const xvc_csb_motion &m = p_.motions[mi];
    const int n = m.state;
    const xvc_cs_state &s = states_[n];
    const bool s_lic = (s.flags & XVC_CS_STATE_LIC) != 0;
    const bool is_motion = s.kind == XVC_CS_INTER || s.kind == XVC_CS_MOTION;
    if (!is_motion || !s.supported) continue;
    if (s_lic && !(m.nb >= 0 && p_.lic_folds)) continue;   // the serial steps (motion_lic)
    folded_[n] = 1;
    const xvc_csb_neighbours *nbr = s_lic ? &p_.nb[m.nb] : nullptr;
    pass_first_[n] = (int64_t)passes_.size();
    int me_next = s.me_first, aff_next = s.aff_first;
    for (int which = 0; which < 2; which++) {
      const xvc_csb_pass_in &pin = which ? m.affine : m.plain;
      if (!pin.n) continue;
      if (pin.n != p_.n_ref[0] + p_.n_ref[1]) return -2;
      if (s_lic && which) return -3;      // (never together: inter_search.cc:215-219)
      xvcgpu_cs_pass p = Zero<xvcgpu_cs_pass>();
      p.x = s.x; p.y = s.y; p.w = s.w; p.h = s.h;
      p.flags = (uint8_t)((pin.fullpel & 1) | (which ? XVC_CS_AFFINE : 0) | (s_lic ? XVC_CS_LIC : 0));
      p.num_refs[0] = (uint8_t)p_.n_ref[0];
      p.num_refs[1] = (uint8_t)p_.n_ref[1];
      for (int r = 0; r < R3; r++) p.same_poc_in_l0[r] = (int8_t)same[r];
      p.lambda16 = pin.lambda16;
      p.ictx = pin.ictx;
      for (int l = 0; l < 2; l++)
        for (int r = 0; r < R3; r++) p.uni_job[l][r] = p.start_dist[l][r] = p.prev_job[l][r] = -1;
      p.plain_pass = -1;
      p.eval = -1;
      p.bi_iterations = 1;
      const int pi = (int)passes_.size();
      const xvc_csb_ref_entry *en = p_.entries + pin.first;
      for (int k = 0; k < pin.n; k++) {
        const int l = en[k].list, r = en[k].ref_idx;
        if (l < 0 || l > 1 || r < 0 || r >= p_.n_ref[l]) return -4;
        std::memcpy(p.mvp[l][r], en[k].mvp, sizeof(en[k].mvp));
        p.slot[l][r] = (int8_t)slot_of(p_.ref_poc[l][r]);
      }
      auto start_into_slots = [&](int iflags) -> int {
        // two predictions per (list, picture) into the scratch slots + their SAD against the
        // original: EvalStartMvp of a LIC CU compares COMPENSATED predictions
        // (MotionCompensationMv(..., post_filter = true), inter_search.cc:980), the affine
        // pass's start predicts with the three corner vectors
        const int first = (int)aff_start_inter_.size();
        for (int k = 0; k < pin.n; k++) {
          const int l = en[k].list, r = en[k].ref_idx;
          p.start_dist[l][r] = n_start_dist;
          for (int cand = 0; cand < 2; cand++) {
            const int kk = (int)aff_start_inter_.size() - first;
            xvcgpu_inter_block ib = Zero<xvcgpu_inter_block>();
            ib.x = s.x; ib.y = s.y; ib.w = s.w; ib.h = s.h;
            ib.flags = (uint8_t)iflags;
            if (nbr) {
              ib.neighbors = (uint8_t)(nbr->has_above * 1 + nbr->has_left * 2);
              ib.above_x = nbr->above_x; ib.above_y = nbr->above_y;
              ib.left_x = nbr->left_x; ib.left_y = nbr->left_y;
            }
            ib.ref[0] = p.slot[l][r];
            ib.ref[1] = -1;
            if (iflags == XVC_INTER_LIC) {
              ib.mv[0][0][0] = p.mvp[l][r][cand][0][0];
              ib.mv[0][0][1] = p.mvp[l][r][cand][0][1];
            } else {
              std::memcpy(ib.mv[0], p.mvp[l][r][cand], sizeof(ib.mv[0]));
            }
            aff_start_inter_.push_back(ib);
            xvcgpu_block_pos d = {(int16_t)(kSlot * kk), 0};
            aff_start_dst_.push_back(d);
            xvcgpu_metric_cand c = {(int16_t)(kSlot * kk), 0, s.w, s.h, 3, 0, 0, 0};
            aff_start_cands_.push_back(c);
            xvcgpu_copy_block cp = {s.x, s.y, (int16_t)(kSlot * kk), 0, s.w, s.h, 0, 0};
            aff_start_copy_.push_back(cp);
            n_start_dist++;
          }
        }
        const int cnt = (int)aff_start_inter_.size() - first;
        if (cnt > kMaxSlots) return -5;
        AffStart as = {first, cnt, p.start_dist[en[0].list][en[0].ref_idx]};
        aff_start_[pi] = as;
        return 0;
      };
      auto plain_uni = [&]() -> int {
        std::vector<SlotJob> ug;
        for (int k = 0; k < pin.n; k++) {
          if (en[k].reused) continue;
          const int l = en[k].list, r = en[k].ref_idx;
          if (me_next >= p_.n_me || p_.me_ref[me_next] != p.slot[l][r]) return -6;
          p.uni_job[l][r] = me_next;
          me_work_[me_next].mvp_x = me_work_[me_next].mvp_y = kDevice;
          ug.push_back(SlotJob(p.slot[l][r], me_next));
          me_next++;
        }
        uni_groups_[pi] = ug;
        p.bi_job = n_bi_slots;
        return 0;
      };
      int rc = 0;
      if (which == 0 && s_lic) {
        if ((rc = start_into_slots(XVC_INTER_LIC))) return rc;
        if ((rc = plain_uni())) return rc;
        xvcgpu_mc_lic_block q = Zero<xvcgpu_mc_lic_block>();
        q.x = s.x; q.y = s.y; q.w = s.w; q.h = s.h;
        q.neighbors = (uint8_t)(nbr->has_above * 1 + nbr->has_left * 2);
        q.above_x = nbr->above_x; q.above_y = nbr->above_y;
        q.left_x = nbr->left_x; q.left_y = nbr->left_y;
        bi_lic_rows[n_bi_slots] = q;
        n_bi_slots += kBiSlots;
      } else if (which == 0) {
        // EvalStartMvp: two luma predictions + SAD per (list, picture), re-used ones too,
        // grouped by picture slot
        std::vector<int> order(pin.n);
        for (int k = 0; k < pin.n; k++) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
          return p.slot[en[a].list][en[a].ref_idx] < p.slot[en[b].list][en[b].ref_idx];
        });
        std::vector<StartGroup> groups;
        for (int oi = 0; oi < pin.n; oi++) {
          const int l = en[order[oi]].list, r = en[order[oi]].ref_idx;
          const int sl = p.slot[l][r];
          p.start_dist[l][r] = n_start_dist;
          for (int cand = 0; cand < 2; cand++) {
            xvcgpu_mc_metric_cand c = {s.x, s.y, s.w, s.h, 3, 0, p.mvp[l][r][cand][0][0],
                                       p.mvp[l][r][cand][0][1]};
            start_cands_.push_back(c);
          }
          if (!groups.empty() && groups.back().slot == sl) {
            groups.back().count += 2;
          } else {
            StartGroup g = {sl, n_start_dist, 2, (int)start_cands_.size() - 2};
            groups.push_back(g);
          }
          n_start_dist += 2;
        }
        start_groups_[pi] = groups;
        if ((rc = plain_uni())) return rc;
        n_bi_slots += kBiSlots;
      } else {
        p.plain_pass = pi - 1;
        if ((rc = start_into_slots(XVC_INTER_AFFINE))) return rc;
        std::vector<SlotJob> ug;
        for (int k = 0; k < pin.n; k++) {
          if (en[k].reused) continue;
          const int l = en[k].list, r = en[k].ref_idx;
          if (aff_next >= p_.n_aff || p_.aff_ref[2 * aff_next] != p.slot[l][r]) return -7;
          p.uni_job[l][r] = (int32_t)aff_rows.size();
          ug.push_back(SlotJob(p.slot[l][r], (int)aff_rows.size()));
          aff_rows.push_back(aff_next);
          aff_next++;
        }
        aff_uni_groups_[pi] = ug;
        p.bi_job = (int32_t)aff_rows.size();
        aff_rows.insert(aff_rows.end(), kBiSlots, -1);
      }
      passes_.push_back(p);
    }
    if (m.plain.n && me_next != s.me_first + s.me_count) return -8;
    if (m.affine.n && aff_next != s.aff_first + s.aff_uni_count) return -9;
    pass_count_[n] = (int64_t)passes_.size() - pass_first_[n];
    if (s.kind == XVC_CS_INTER && pass_count_[n]) passes_.back().eval = s.ev;
  }
  bi_lic_work_.assign(std::max(n_bi_slots, 1), Zero<xvcgpu_mc_lic_block>());
  for (std::map<int, xvcgpu_mc_lic_block>::const_iterator it = bi_lic_rows.begin();
       it != bi_lic_rows.end(); ++it)
    for (int k = 0; k < kBiSlots; k++) bi_lic_work_[it->first + k] = it->second;
  aff_work_.assign(aff_rows.size(), Zero<xvcgpu_affine_me_block>());
  aff_work_src_ = aff_rows;
  for (size_t i = 0; i < aff_rows.size(); i++)
    if (aff_rows[i] >= 0) {
      aff_work_[i] = p_.aff_jobs[aff_rows[i]];
      for (int c = 0; c < 3; c++)
        for (int k = 0; k < 2; k++) aff_work_[i].mvp[c][k] = aff_work_[i].bootstrap[c][k] = kDevice;
    }
  // the *_refs forms: per job the slot(s) of the picture(s) it works on (255: no job)
  start_slots_.assign(start_cands_.size(), 255);
  for (std::map<int, std::vector<StartGroup> >::const_iterator it = start_groups_.begin();
       it != start_groups_.end(); ++it)
    for (size_t g = 0; g < it->second.size(); g++)
      for (int k = 0; k < it->second[g].count; k++)
        start_slots_[it->second[g].first_cand + k] = (uint8_t)it->second[g].slot;
  me_slots_.resize(p_.n_me);
  for (int i = 0; i < p_.n_me; i++) me_slots_[i] = (uint8_t)p_.me_ref[i];
  bi_slots_.assign(2 * (size_t)std::max(n_bi_slots, 1), 255);
  aff_slots_.assign(2 * std::max(aff_work_.size(), (size_t)1), 255);
  for (size_t pi = 0; pi < passes_.size(); pi++) {
    const xvcgpu_cs_pass &p = passes_[pi];
    const bool affine = (p.flags & XVC_CS_AFFINE) != 0;
    if (affine) {
      const std::vector<SlotJob> &ug = aff_uni_groups_[(int)pi];
      for (size_t k = 0; k < ug.size(); k++)
        aff_slots_[2 * ug[k].second] = aff_slots_[2 * ug[k].second + 1] = (uint8_t)ug[k].first;
    }
    if (p.num_refs[1]) {
      for (int sl = 0; sl < 2; sl++)
        for (int r = 0; r < p.num_refs[sl]; r++)
          for (int o = 0; o < p.num_refs[1 - sl]; o++) {
            const int k = p.bi_job + (sl * R3 + r) * R3 + o;
            std::vector<uint8_t> &dst = affine ? aff_slots_ : bi_slots_;
            dst[2 * k] = (uint8_t)p.slot[sl][r];
            dst[2 * k + 1] = (uint8_t)p.slot[1 - sl][o];
          }
    }
  }
  // the evaluations' prediction jobs: motion composed on the device for inter states
  ev_inter_work_.assign(p_.ev_inter, p_.ev_inter + 3 * (size_t)p_.n_ev);
  for (int n = 0; n < n_st; n++) {
    if (states_[n].kind != XVC_CS_INTER || !folded_[n]) continue;
    const int e = states_[n].ev;
    for (int c = 0; c < 3; c++) {
      xvcgpu_inter_block &b = ev_inter_work_[3 * e + c];
      // overwritten by the fold (a LIC CU's neighbour fields stay: they are the caller's)
      b.ref[0] = 0; b.ref[1] = -1;
      b.flags = 0;
      for (int l = 0; l < 2; l++)
        for (int k = 0; k < 3; k++) b.mv[l][k][0] = b.mv[l][k][1] = 12345;
    }
  }
  return 0;
}

// The records of xvcgpu_cs_merge_fold for every merge ranking and the evaluation slots it
// fills: four per ranking, the Y, U, V prediction jobs with the CU's geometry, the motion
// left to the fold (SearchMergeCandidates inter_search.cc:165-197, CompressMerge
// cu_encoder.cc:598-628).
void CuStateBuilder::BuildMergeFolds() {
  const int n_m = p_.n_merges;
  mg_fold_.assign(n_m, Zero<xvcgpu_cs_merge>());
  mg_slots_.assign((size_t)n_m * kMergeSlots * 3, Zero<xvcgpu_inter_block>());
  merge_state_.assign(n_m, -1);
  for (int m = 0; m < n_m; m++) {
    const xvc_csb_merge &g = p_.merges[m];
    mg_fold_[m].lambda_sqrt = g.lambda_sqrt;
    mg_fold_[m].dist = mg_fold_[m].cand = 5 * m;
    mg_fold_[m].slot = kMergeSlots * m;
    const bool has = g.any_lic && g.nb >= 0;
    for (int sl = 0; sl < kMergeSlots; sl++)
      for (int c = 0; c < 3; c++) {
        xvcgpu_inter_block &b = mg_slots_[((size_t)m * kMergeSlots + sl) * 3 + c];
        b.x = g.x; b.y = g.y; b.w = g.w; b.h = g.h;
        b.comp = (uint8_t)c;
        b.ref[0] = b.ref[1] = -1;           // the fold's to write (and XVC_INTER_LIC)
        for (int l = 0; l < 2; l++)
          for (int k = 0; k < 3; k++) b.mv[l][k][0] = b.mv[l][k][1] = kDevice;
        if (has) {    // what a LIC candidate's prediction reads besides its motion
          const xvc_csb_neighbours &nb = p_.nb[g.nb];
          b.neighbors = (uint8_t)(nb.has_above * 1 + nb.has_left * 2);
          b.above_x = nb.above_x; b.above_y = nb.above_y;
          b.left_x = nb.left_x; b.left_y = nb.left_y;
        }
        b.flags = 0;
      }
    if (g.state >= 0) merge_state_[m] = g.state;
  }
}

// Per evaluation state one block of distortion candidates: [3 cbf-zero (Y, U, V)] [one per
// TransformAndReconstruct call]; the merge rankings' and the start predictors' distortions
// as candidates against the original picture; per call where its original and prediction lie.
void CuStateBuilder::BuildEvalCands() {
  const int n_st = p_.n_states;
  edist_first_.assign(n_st, -1);
  n_edist = 0;
  call_comp_.assign(p_.call_comp, p_.call_comp + p_.n_calls);
  for (int ns = 0; ns < n_st; ns++) {
    const xvc_cs_state &r = states_[ns];
    if (r.ev < 0) continue;
    const xvc_csb_eval &ev = p_.evals[r.ev];
    const int cf = r.call_first, k = r.call_pass0 + r.call_pass1;
    edist_first_[ns] = n_edist;
    for (int c = 0; c < 3; c++) {
      xvcgpu_eval_cand b = Zero<xvcgpu_eval_cand>();
      b.x = ev.dz[c].x; b.y = ev.dz[c].y; b.w = ev.dz[c].w; b.h = ev.dz[c].h;
      b.metric = ev.dz[c].metric; b.qp = ev.dz[c].qp;
      b.comp = (uint8_t)c;
      b.versus = 0;
      b.ox = (int16_t)(ev.x >> (c ? 1 : 0));
      b.oy = (int16_t)(ev.y >> (c ? 1 : 0));
      b.orig_at = 1;
      b.weight = ev.weight[c];
      ev_cands_.push_back(b);
    }
    for (int i = cf; i < cf + k; i++) {
      const xvcgpu_metric_cand &cc = p_.call_cand[i];
      const int comp = p_.call_comp[i];
      xvcgpu_eval_cand b = Zero<xvcgpu_eval_cand>();
      b.x = cc.x; b.y = cc.y; b.w = cc.w; b.h = cc.h; b.metric = cc.metric; b.qp = cc.qp;
      b.comp = (uint8_t)comp;
      b.versus = 1;
      b.ox = (int16_t)(ev.x >> (comp != 0));
      b.oy = (int16_t)(ev.y >> (comp != 0));
      b.orig_at = 1;
      b.weight = ev.weight[comp];
      ev_cands_.push_back(b);
    }
    n_edist += 3 + k;
  }
  ev_cands_copy_ = ev_cands_;
  for (size_t i = 0; i < ev_cands_copy_.size(); i++) ev_cands_copy_[i].orig_at = 0;
  call_pos_.assign(2 * (size_t)p_.n_calls, Zero<xvcgpu_block_pos>());
  for (int i = 0; i < p_.n_calls; i++) {
    const xvc_csb_eval &ev = p_.evals[p_.call_ev[i]];
    const int sh = p_.call_comp[i] != 0;
    call_pos_[2 * i].x = (int16_t)(ev.x >> sh);
    call_pos_[2 * i].y = (int16_t)(ev.y >> sh);
  }
  mg_ecands_.assign(5 * (size_t)p_.n_merges, Zero<xvcgpu_eval_cand>());
  for (int m = 0; m < p_.n_merges; m++)
    for (int c = 0; c < 5; c++) {
      const xvcgpu_metric_cand &mc = p_.mg_cands[5 * m + c];
      xvcgpu_eval_cand &b = mg_ecands_[5 * m + c];
      b.x = mc.x; b.y = mc.y; b.w = mc.w; b.h = mc.h; b.metric = mc.metric;
      b.ox = p_.merges[m].x; b.oy = p_.merges[m].y;
      b.orig_at = 1;
      b.weight = 1.0;
    }
  aff_start_ecands_.assign(aff_start_cands_.size(), Zero<xvcgpu_eval_cand>());
  for (size_t i = 0; i < aff_start_cands_.size(); i++) {
    const xvcgpu_metric_cand &mc = aff_start_cands_[i];
    xvcgpu_eval_cand &b = aff_start_ecands_[i];
    b.x = mc.x; b.y = mc.y; b.w = mc.w; b.h = mc.h; b.metric = mc.metric;
    b.ox = aff_start_copy_[i].sx; b.oy = aff_start_copy_[i].sy;
    b.orig_at = 1;
    b.weight = 1.0;
  }
}

const void *CuStateBuilder::Array(int which, int64_t *bytes) const {
#define ARR(v) do { *bytes = (int64_t)((v).size() * sizeof((v)[0])); return (v).empty() ? nullptr : (const void *)&(v)[0]; } while (0)
  switch (which) {
    case XVC_CSB_PASSES: ARR(passes_);
    case XVC_CSB_PASS_FIRST: ARR(pass_first_);
    case XVC_CSB_PASS_COUNT: ARR(pass_count_);
    case XVC_CSB_FOLDED: ARR(folded_);
    case XVC_CSB_START_CANDS: ARR(start_cands_);
    case XVC_CSB_START_SLOTS: ARR(start_slots_);
    case XVC_CSB_AFF_START_INTER: ARR(aff_start_inter_);
    case XVC_CSB_AFF_START_DST: ARR(aff_start_dst_);
    case XVC_CSB_AFF_START_CANDS: ARR(aff_start_cands_);
    case XVC_CSB_AFF_START_COPY: ARR(aff_start_copy_);
    case XVC_CSB_ME_WORK: ARR(me_work_);
    case XVC_CSB_BI_LIC_WORK: ARR(bi_lic_work_);
    case XVC_CSB_AFF_WORK: ARR(aff_work_);
    case XVC_CSB_AFF_WORK_SRC: ARR(aff_work_src_);
    case XVC_CSB_ME_SLOTS: ARR(me_slots_);
    case XVC_CSB_BI_SLOTS: ARR(bi_slots_);
    case XVC_CSB_AFF_SLOTS: ARR(aff_slots_);
    case XVC_CSB_EV_INTER_WORK: ARR(ev_inter_work_);
    case XVC_CSB_MG_FOLD: ARR(mg_fold_);
    case XVC_CSB_MG_SLOTS: ARR(mg_slots_);
    case XVC_CSB_MERGE_STATE: ARR(merge_state_);
    case XVC_CSB_EV_CANDS: ARR(ev_cands_);
    case XVC_CSB_EV_CANDS_COPY: ARR(ev_cands_copy_);
    case XVC_CSB_EDIST_FIRST: ARR(edist_first_);
    case XVC_CSB_CALL_POS: ARR(call_pos_);
    case XVC_CSB_MG_ECANDS: ARR(mg_ecands_);
    case XVC_CSB_AFF_START_ECANDS: ARR(aff_start_ecands_);
    default: *bytes = 0; return nullptr;
  }
#undef ARR
}

// The op program of the states [first, first + n): one chain (ending in a SYNC) per state,
// or per visit of a CU position (consecutive states of one CU), or - XVC_CSB_LIVE - the
// chains a live encoder could issue: a chain ends wherever the reference's control reads a
// cost that needs the host's entropy coder (GetCuCostWithoutSplit, cu_encoder.cc:431-515):
// after every CompressInter's evaluation and inside it in front of the gated second
// transform pass (cost_full > best_cu_cost * 1.1, inter_search.cc:347-361); what the device
// folds decide needs no wait, and a merge ranking and its candidates' evaluations are one
// chain.  XVC_CSB_REFS_FORM: a step of SearchMotion into all the CU's reference pictures as
// one launch, the read-backs of a chain merged where their ranges touch.
const xvc_cs_op *CuStateBuilder::Program(const xvc_csb_addrs &a, const xvc_csb_intra &in,
                                         int first, int n, int flags, int64_t *n_ops) {
  const bool by_position = (flags & XVC_CSB_BY_POSITION) != 0, verify = (flags & XVC_CSB_VERIFY) != 0;
  const bool refs_form = (flags & XVC_CSB_REFS_FORM) != 0, live = (flags & XVC_CSB_LIVE) != 0;
  const bool no_copies = (flags & XVC_CSB_NO_COPIES) != 0, fused_eval = (flags & XVC_CSB_FUSED_EVAL) != 0;
  const bool merge_fold = (flags & XVC_CSB_MERGE_FOLD) != 0;
  ops_.clear();
  const uint64_t I_me = sizeof(xvcgpu_me_block), I_res = sizeof(xvcgpu_me_result),
                 I_bi = sizeof(xvcgpu_bi_block), I_aff = sizeof(xvcgpu_affine_me_block),
                 I_affr = sizeof(xvcgpu_affine_me_result), I_mcm = sizeof(xvcgpu_mc_metric_cand),
                 I_inter = sizeof(xvcgpu_inter_block), I_pos = sizeof(xvcgpu_block_pos),
                 I_cand = sizeof(xvcgpu_metric_cand), I_copy = sizeof(xvcgpu_copy_block),
                 I_tx = sizeof(xvcgpu_tx_block), I_prm = sizeof(xvcgpu_rdoq_params),
                 I_ctx = sizeof(xvcgpu_rdoq_contexts), I_result = sizeof(xvcgpu_cs_result),
                 I_intra = sizeof(xvcgpu_intra_block);
  auto op = [&](int code, int n_, int r0, int r1, int i0, double f, std::initializer_list<uint64_t> p) {
    xvc_cs_op o;
    std::memset(&o, 0, sizeof(o));
    o.opcode = code; o.n = n_; o.r0 = r0; o.r1 = r1; o.i0 = i0; o.f = f;
    int k = 0;
    for (uint64_t v : p) o.p[k++] = v;
    ops_.push_back(o);
  };
  struct Fetch { uint64_t dev, host; int64_t nb; };
  std::vector<Fetch> pending;
  auto fetch_now = [&](uint64_t dev, uint64_t host, int64_t nb) {
    if (nb) op(XVC_OP_FETCH, (int)nb, 0, 0, 0, 0.0, {dev, host});
  };
  auto flush_fetches = [&]() {
    std::stable_sort(pending.begin(), pending.end(), [](const Fetch &x, const Fetch &y) {
      return std::tie(x.dev, x.host, x.nb) < std::tie(y.dev, y.host, y.nb);
    });
    std::vector<Fetch> merged;
    for (size_t i = 0; i < pending.size(); i++) {
      const Fetch &f = pending[i];
      if (!merged.empty() && merged.back().dev + (uint64_t)merged.back().nb >= f.dev &&
          f.host - merged.back().host == f.dev - merged.back().dev) {
        merged.back().nb = std::max(merged.back().nb, (int64_t)(f.dev + f.nb - merged.back().dev));
      } else {
        merged.push_back(f);
      }
    }
    for (size_t i = 0; i < merged.size(); i++) fetch_now(merged[i].dev, merged[i].host, merged[i].nb);
    pending.clear();
  };
  auto fetch = [&](uint64_t dev, uint64_t host, int64_t nb) {
    if (nb <= 0) return;
    if (refs_form) {
      Fetch f = {dev, host, nb};
      pending.push_back(f);
    } else {
      fetch_now(dev, host, nb);
    }
  };
  auto stage = [&](const xvc_cs_state &s) {
    if (s.nb_count)   // the reconstruction of that moment around a LIC state's CU
      op(XVC_OP_COPY, s.nb_count, PIC_NB, PIC_REC, 0, 0.0, {a.d_nb_copy + (uint64_t)s.nb_first * I_copy});
  };
  // A LIC state's SearchMotion in the serial form (no device folds): AC-only searches, then
  // the refinement against the compensated prediction of the other list.
  auto motion_lic = [&](const xvc_cs_state &s) {
    const int ms = std::max((int)s.w, (int)s.h);
    const int mf = s.me_first, mc = s.me_count;
    for (int j = mf; j < mf + mc; j++)
      op(XVC_OP_ME, 1, p_.me_ref[j], 1, ms, 0.0, {a.d_me + j * I_me, a.d_me_res + j * I_res});
    fetch(a.d_me_res + mf * I_res, a.h_me_res + mf * I_res, (int64_t)mc * (int64_t)I_res);
    const int bf = s.bi_first, bc = s.bi_count;
    if (bc && live) {   // the host's fold over lists and pictures picks the bootstrap
      flush_fetches();
      op(XVC_OP_SYNC, 0, s.kind, 0, 0, 0.0, {});
    }
    for (int j = bf; j < bf + bc; j++)
      op(XVC_OP_BI_LIC, 1, in.bi_ref[2 * j], in.bi_ref[2 * j + 1], ms, 0.0,
         {a.d_bi + j * I_bi, a.d_bi_res + j * I_res, a.d_bi_lic + (uint64_t)j * 24});
    fetch(a.d_bi_res + bf * I_res, a.h_bi_res + bf * I_res, (int64_t)bc * (int64_t)I_res);
    if (live && s.kind == XVC_CS_INTER) {   // the three-way choice is the host's
      flush_fetches();
      op(XVC_OP_SYNC, 0, s.kind, 0, 0, 0.0, {});
    }
  };
  // CompressIntra: the SATD pre-selection, a wait (the host sorts with the mode bits), then
  // the kept modes' PredictAndTransform alternatives one behind the other at the CU's place
  // (intra_search.cc:61-82, :118-150 decide nothing between the modes); a live chain waits
  // behind the luma modes and behind the chroma modes.
  auto intra = [&](const xvc_cs_state &s) {
    const int ms = std::max((int)s.w, (int)s.h);
    const int k = s.in_satd;
    if (k >= 0) {
      op(XVC_OP_INTRA_SATD, 1, 0, 0, ms, 0.0,
         {a.d_in_satd_jobs + (uint64_t)k * I_intra, a.d_in_satd + 4ull * 67 * k});
      fetch(a.d_in_satd + 4ull * 67 * k, a.h_in_satd + 4ull * 67 * k, 4 * 67);
      flush_fetches();
      op(XVC_OP_SYNC, 0, XVC_CS_INTRA, 0, 0, 0.0, {});
    }
    const int ia = s.in_first, nc = s.in_count;
    for (int c = ia; c < ia + nc; c++) {
      const int sf = in.in_stage[2 * c], sc = in.in_stage[2 * c + 1];
      if (sc) op(XVC_OP_COPY, sc, PIC_NB, PIC_REC, 0, 0.0, {a.d_nb_copy + (uint64_t)sf * I_copy});
      op(XVC_OP_INTRA_PRED, 1, 0, 0, 0, 0.0, {a.d_in_pred + (uint64_t)c * I_intra});
      op(XVC_OP_RESIDUAL_INTRA, 1, 0, 0, 0, 0.0,
         {a.d_in_tx + (uint64_t)c * I_tx, a.d_in_off + 4ull * c, a.d_in_nnz + 4ull * c,
          a.d_in_contexts + (uint64_t)in.in_ctx[c] * I_ctx, a.d_in_prm + (uint64_t)c * I_prm});
      op(XVC_OP_METRIC, 1, PIC_ORIG, PIC_IREC, in.in_comp[c], in.in_weight[c],
         {a.d_in_cand + (uint64_t)c * I_cand, a.d_in_dist + 8ull * c});
      if (live && c + 1 < ia + nc && in.in_comp[c] == 0 && in.in_comp[c + 1] != 0) {
        fetch(a.d_in_nnz + 4ull * ia, a.h_in_nnz + 4ull * ia, 4 * (int64_t)(c + 1 - ia));
        fetch(a.d_in_dist + 8ull * ia, a.h_in_dist + 8ull * ia, 8 * (int64_t)(c + 1 - ia));
        flush_fetches();
        op(XVC_OP_SYNC, 0, XVC_CS_INTRA, 0, 0, 0.0, {});
      }
    }
    fetch(a.d_in_nnz + 4ull * ia, a.h_in_nnz + 4ull * ia, 4 * (int64_t)nc);
    fetch(a.d_in_dist + 8ull * ia, a.h_in_dist + 8ull * ia, 8 * (int64_t)nc);
    if (nc) {
      const int64_t la = in.in_off[ia];
      const int64_t lb = ia + nc < in.n_in ? (int64_t)in.in_off[ia + nc] : in.n_in_levels;
      fetch(a.d_in_levels + 2ull * la, a.h_in_levels + 2ull * la, 2 * (lb - la));
    }
  };
  auto motion = [&](const xvc_cs_state &s, int n_state) {
    if ((s.flags & XVC_CS_STATE_LIC) && !folded_[n_state]) {
      motion_lic(s);
      return;
    }
    const int ms = std::max((int)s.w, (int)s.h);
    const int cls = ms <= 16 ? 16 : (ms <= 32 ? 32 : 64);
    const int pf = (int)pass_first_[n_state], pc = (int)pass_count_[n_state];
    for (int pi = pf; pi < pf + pc; pi++) {
      const xvcgpu_cs_pass &p = passes_[pi];
      const bool affine = (p.flags & XVC_CS_AFFINE) != 0;
      const bool licp = (p.flags & XVC_CS_LIC) != 0;   // XVC_INTER_LIC start predictions, LIC searches per picture
      const uint64_t P = a.passes;                     // the folds index the arrays absolutely (i0 = pass)
      if (licp || affine) {
        const AffStart &as = aff_start_[pi];
        const uint64_t inter = a.aff_start_inter + as.first * I_inter, dst = a.aff_start_dst + as.first * I_pos;
        if (licp) {
          op(XVC_OP_INTER_PRED, as.n, 1, PIC_S_PRED, 0, 0.0, {inter, dst});
        } else {
          if (!no_copies)
            op(XVC_OP_COPY, as.n, PIC_ORIG, PIC_S_ORIG, 0, 0.0, {a.aff_start_copy + as.first * I_copy});
          op(XVC_OP_INTER_PRED, as.n, 0, PIC_S_PRED, 0, 0.0, {inter, dst});
        }
        if (no_copies) {
          op(XVC_OP_EVAL_DIST, as.n, 1, 0, 0, 0.0,
             {a.aff_start_ecands + (uint64_t)as.first * 24, a.start_dist + 8ull * as.start_dist});
        } else {
          if (licp) op(XVC_OP_COPY, as.n, PIC_ORIG, PIC_S_ORIG, 0, 0.0, {a.aff_start_copy + as.first * I_copy});
          op(XVC_OP_METRIC, as.n, PIC_S_ORIG, PIC_S_PRED, 0, 1.0,
             {a.aff_start_cands + as.first * I_cand, a.start_dist + 8ull * as.start_dist});
        }
      } else if (refs_form) {
        const std::vector<StartGroup> &g = start_groups_[pi];
        int kk = 0;
        for (size_t q = 0; q < g.size(); q++) kk += g[q].count;
        op(XVC_OP_MC_METRIC_REFS, kk, 0, 0, 0, 0.0,
           {a.start_cands + g[0].first_cand * I_mcm, a.start_dist + 8ull * g[0].first_dist,
            a.start_slots + (uint64_t)g[0].first_cand});
      } else {
        const std::vector<StartGroup> &g = start_groups_[pi];
        for (size_t q = 0; q < g.size(); q++)
          op(XVC_OP_MC_METRIC, g[q].count, g[q].slot, 0, 0, 0.0,
             {a.start_cands + g[q].first_cand * I_mcm, a.start_dist + 8ull * g[q].first_dist});
      }
      op(XVC_OP_START_FOLD, 1, 0, 0, pi, 0.0, {P, a.start_dist, a.me_work, a.me_res_c, a.aff_work});
      const std::vector<SlotJob> &ug = affine ? aff_uni_groups_[pi] : uni_groups_[pi];
      if (licp) {
        for (size_t q = 0; q < ug.size(); q++)
          op(XVC_OP_ME, 1, ug[q].first, 1, ms, 0.0,
             {a.me_work + ug[q].second * I_me, a.me_res_c + ug[q].second * I_res});
      } else if (refs_form && !ug.empty()) {
        const int j0 = ug[0].second;
        if (!affine)
          op(XVC_OP_ME_REFS, (int)ug.size(), 0, 0, cls, 0.0,
             {a.me_work + j0 * I_me, a.me_res_c + j0 * I_res, a.me_slots + (uint64_t)j0});
        else
          op(XVC_OP_AFFINE_REFS, (int)ug.size(), 0, 0, s.h, 0.0,
             {a.aff_work + j0 * I_aff, a.aff_res_c + j0 * I_affr, a.aff_slots + 2ull * j0});
      } else if (!affine) {
        for (size_t q = 0; q < ug.size(); q++)
          op(XVC_OP_ME, 1, ug[q].first, 0, ms, 0.0,
             {a.me_work + ug[q].second * I_me, a.me_res_c + ug[q].second * I_res});
      } else {
        for (size_t q = 0; q < ug.size(); q++)
          op(XVC_OP_AFFINE, 1, ug[q].first, ug[q].first, 0, 0.0,
             {a.aff_work + ug[q].second * I_aff, a.aff_res_c + ug[q].second * I_affr});
      }
      op(XVC_OP_UNI_FOLD, 1, 0, 0, pi, 0.0, {P, a.me_res_c, a.aff_res_c, a.bi_work, a.aff_work});
      if (p.num_refs[1] && (licp || !refs_form)) {
        const int bj = p.bi_job;
        for (int sl = 0; sl < 2; sl++)
          for (int r = 0; r < p.num_refs[sl]; r++)
            for (int o = 0; o < p.num_refs[1 - sl]; o++) {
              const uint64_t k = (uint64_t)(bj + (sl * R3 + r) * R3 + o);
              const int rs = p.slot[sl][r], ro = p.slot[1 - sl][o];
              if (licp)
                op(XVC_OP_BI_LIC, 1, rs, ro, ms, 0.0,
                   {a.bi_work + k * I_bi, a.bi_res_c + k * I_res, a.bi_lic_work + k * 24});
              else if (!affine)
                op(XVC_OP_BI, 1, rs, ro, ms, 0.0, {a.bi_work + k * I_bi, a.bi_res_c + k * I_res});
              else
                op(XVC_OP_AFFINE, 1, rs, ro, 0, 0.0, {a.aff_work + k * I_aff, a.aff_res_c + k * I_affr});
            }
      } else if (p.num_refs[1]) {
        const uint64_t bj = (uint64_t)p.bi_job;
        if (!affine)
          op(XVC_OP_BI_REFS, kBiSlots, 0, 0, cls, 0.0,
             {a.bi_work + bj * I_bi, a.bi_res_c + bj * I_res, a.bi_slots + 2 * bj});
        else
          op(XVC_OP_AFFINE_REFS, kBiSlots, 0, 0, s.h, 0.0,
             {a.aff_work + bj * I_aff, a.aff_res_c + bj * I_affr, a.aff_slots + 2 * bj});
      }
      op(XVC_OP_BI_FOLD, 1, 0, 0, pi, 0.0, {P, a.bi_res_c, a.aff_res_c, a.ev_inter_work});
    }
    fetch(a.results + pf * I_result, a.h_results + pf * I_result, (int64_t)pc * (int64_t)I_result);
  };
  // the evaluation's slot, when its ranking is folded by THIS program
  auto slot_of = [&](int e) {
    int sl = merge_fold ? p_.evals[e].merge_slot : -1;
    if (sl >= 0) {
      const int64_t ms = merge_state_[sl / kMergeSlots];
      if (!(first <= ms && ms < first + n)) sl = -1;
    }
    return sl;
  };
  auto evaluation = [&](const xvc_cs_state &s, int n_state) {
    const int lic = (s.flags & XVC_CS_STATE_LIC) ? 1 : 0;   // INTER_PRED: neighbours from the reconstruction
    const uint64_t e = (uint64_t)s.ev;
    const int n0 = s.call_pass0, n1 = s.call_pass1;
    const uint64_t cf = (uint64_t)s.call_first;
    const int k = n0 + n1;
    const uint64_t ed = (uint64_t)edist_first_[n_state];
    const int sl = slot_of(s.ev);
    const uint64_t pred_jobs = sl >= 0 ? a.mg_slots + 3ull * sl * I_inter : a.ev_inter_work + 3 * e * I_inter;
    const uint64_t ecands = no_copies ? a.ev_cands : a.ev_cands_copy;
    const uint64_t ctx = a.d_contexts + (uint64_t)p_.ev_ctx[s.ev] * I_ctx;
    const bool fe = no_copies && fused_eval;
    auto residual = [&](uint64_t c0, int cnt, uint64_t e0, int head) {
      if (fe)
        op(XVC_OP_RESIDUAL, cnt, head, 0, 0, 0.0,
           {a.d_call_tx + c0 * I_tx, a.d_call_off + 4 * c0, a.z_nnz + 4 * c0, ctx,
            a.d_call_prm + c0 * I_prm, a.call_pos + 2 * c0 * I_pos, ecands + e0 * 24, a.z_edist + 8 * e0});
      else
        op(XVC_OP_RESIDUAL, cnt, 0, 0, 0, 0.0,
           {a.d_call_tx + c0 * I_tx, a.d_call_off + 4 * c0, a.z_nnz + 4 * c0, ctx,
            a.d_call_prm + c0 * I_prm, no_copies ? a.call_pos + 2 * c0 * I_pos : 0});
    };
    if (live && n1) {
      // the first transform pass, a wait (the host prices it and decides the gate), then
      // the second pass' calls
      const uint64_t co = (uint64_t)s.copy_first;
      if (!no_copies)
        op(XVC_OP_COPY, 3 + n0, PIC_ORIG, PIC_S_ORIG, 0, 0.0, {a.d_copy_orig + co * I_copy});
      op(XVC_OP_INTER_PRED, 3, lic, PIC_S_PRED, 0, 0.0, {pred_jobs, a.d_ev_dst + 3 * e * I_pos});
      if (!no_copies)
        op(XVC_OP_COPY, n0, PIC_S_PRED, PIC_S_PRED, 0, 0.0, {a.d_call_copy_pred + cf * I_copy});
      residual(cf, n0, ed, 3);
      if (!fe) op(XVC_OP_EVAL_DIST, 3 + n0, no_copies ? 1 : 0, 0, 0, 0.0, {ecands + ed * 24, a.z_edist + 8 * ed});
      flush_fetches();
      op(XVC_OP_SYNC, 0, s.kind, 0, 0, 0.0, {});
      const uint64_t c1 = cf + n0;
      if (!no_copies) {
        op(XVC_OP_COPY, n1, PIC_ORIG, PIC_S_ORIG, 0, 0.0, {a.d_copy_orig + (co + 3 + n0) * I_copy});
        op(XVC_OP_COPY, n1, PIC_S_PRED, PIC_S_PRED, 0, 0.0, {a.d_call_copy_pred + c1 * I_copy});
      }
      residual(c1, n1, ed + 3 + n0, 0);
      if (!fe)
        op(XVC_OP_EVAL_DIST, n1, no_copies ? 1 : 0, 0, 0, 0.0,
           {ecands + (ed + 3 + n0) * 24, a.z_edist + 8 * (ed + 3 + n0)});
      fetch(a.d_levels + 2ull * s.level_first, a.h_levels + 2ull * s.level_first, 2 * s.level_count);
      if (verify && s.kind == XVC_CS_INTER)
        fetch(a.ev_inter_work + 3 * e * I_inter, a.h_ev_inter_out + 3 * e * I_inter, 3 * (int64_t)I_inter);
      if (verify && sl >= 0)
        fetch(pred_jobs, a.z_mg_slots_out + 3ull * sl * I_inter, 3 * (int64_t)I_inter);
      return;
    }
    if (!no_copies)
      op(XVC_OP_COPY, 3 + k, PIC_ORIG, PIC_S_ORIG, 0, 0.0, {a.d_copy_orig + (uint64_t)s.copy_first * I_copy});
    op(XVC_OP_INTER_PRED, 3, lic, PIC_S_PRED, 0, 0.0, {pred_jobs, a.d_ev_dst + 3 * e * I_pos});
    if (verify && sl >= 0)      // the slot's motion, to be held against the capture
      fetch(pred_jobs, a.z_mg_slots_out + 3ull * sl * I_inter, 3 * (int64_t)I_inter);
    if (!no_copies)
      op(XVC_OP_COPY, k, PIC_S_PRED, PIC_S_PRED, 0, 0.0, {a.d_call_copy_pred + cf * I_copy});
    if (fe) {
      // the alternatives' reconstruction and all of the evaluation's distortions (three
      // cbf-zero ones in front) in ONE launch
      op(XVC_OP_RESIDUAL, k, 3, 0, 0, 0.0,
         {a.d_call_tx + cf * I_tx, a.d_call_off + 4 * cf, a.z_nnz + 4 * cf, ctx, a.d_call_prm + cf * I_prm,
          a.call_pos + 2 * cf * I_pos, a.ev_cands + ed * 24, a.z_edist + 8 * ed});
    } else {
      op(XVC_OP_RESIDUAL, k, 0, 0, 0, 0.0,
         {a.d_call_tx + cf * I_tx, a.d_call_off + 4 * cf, a.z_nnz + 4 * cf, ctx, a.d_call_prm + cf * I_prm,
          no_copies ? a.call_pos + 2 * cf * I_pos : 0});
      // the three cbf-zero distortions and every alternative's, one launch
      op(XVC_OP_EVAL_DIST, 3 + k, no_copies ? 1 : 0, 0, 0, 0.0, {ecands + ed * 24, a.z_edist + 8 * ed});
    }
    fetch(a.d_levels + 2ull * s.level_first, a.h_levels + 2ull * s.level_first, 2 * s.level_count);
    if (verify && s.kind == XVC_CS_INTER)   // (an encoder reads the motion from `results`)
      fetch(a.ev_inter_work + 3 * e * I_inter, a.h_ev_inter_out + 3 * e * I_inter, 3 * (int64_t)I_inter);
  };

  int chain_states = 0, chain_kind = 0;
  bool have_prev = false, merge_open = false;   // live: the open chain is a merge ranking + its candidates
  int16_t px = 0, py = 0;
  uint8_t pw = 0, ph = 0;
  for (int n_state = first; n_state < first + n; n_state++) {
    const xvc_cs_state &s = states_[n_state];
    if (!s.supported) continue;
    const bool same_key = have_prev && s.x == px && s.y == py && s.w == pw && s.h == ph;
    bool cut;
    if (live) {
      const bool stay = merge_open && same_key && s.kind == XVC_CS_EVAL && slot_of(s.ev) >= 0;
      merge_open = stay || (s.kind == XVC_CS_MERGE_RANK && merge_fold);
      cut = !stay;
    } else {
      cut = !by_position || !same_key;
    }
    if (chain_states && cut) {
      flush_fetches();
      op(XVC_OP_SYNC, 0, chain_kind, 0, chain_states, 0.0, {});
      chain_states = 0;
    }
    have_prev = true;
    px = s.x; py = s.y; pw = s.w; ph = s.h;
    const int kind = s.kind;
    stage(s);
    if (kind == XVC_CS_MERGE_RANK) {
      const uint64_t m = (uint64_t)s.merge * 5;
      if (!no_copies) op(XVC_OP_COPY, 5, PIC_ORIG, PIC_S_ORIG, 0, 0.0, {a.d_mg_copy + m * I_copy});
      op(XVC_OP_INTER_PRED, 5, (s.flags & XVC_CS_STATE_LIC) ? 1 : 0, PIC_S_PRED, 0, 0.0,
         {a.d_mg_inter + m * I_inter, a.d_mg_dst + m * I_pos});
      if (no_copies)
        op(XVC_OP_EVAL_DIST, 5, 1, 0, 0, 0.0, {a.mg_ecands + m * 24, a.z_mg_dist + 8 * m});
      else
        op(XVC_OP_METRIC, 5, PIC_S_ORIG, PIC_S_PRED, 0, 1.0, {a.d_mg_cands + m * I_cand, a.z_mg_dist + 8 * m});
      if (merge_fold)
        op(XVC_OP_MERGE_FOLD, 1, 0, 0, (int)(m / 5), 0.0,
           {a.mg_fold, a.z_mg_dist, a.d_mg_inter, a.z_mg_res, a.mg_slots});
    } else if (kind == XVC_CS_INTRA) {
      intra(s);
    } else {
      if (kind == XVC_CS_INTER || kind == XVC_CS_MOTION) motion(s, n_state);
      if (kind == XVC_CS_EVAL || kind == XVC_CS_INTER) evaluation(s, n_state);
    }
    chain_kind = chain_states ? std::max(chain_kind, kind) : kind;
    chain_states++;
  }
  if (chain_states) {
    flush_fetches();
    op(XVC_OP_SYNC, 0, chain_kind, 0, chain_states, 0.0, {});
  }
  *n_ops = (int64_t)ops_.size();
  return ops_.empty() ? nullptr : &ops_[0];
}

}  // namespace xvc_gpu

struct xvc_csb {
  xvc_gpu::CuStateBuilder b;
};

extern "C" {

int xvc_host_csb_build(const xvc_csb_picture *pic, xvc_csb **out) {
  if (!pic || !out) return -1;
  *out = nullptr;
  xvc_csb *h = new xvc_csb();
  const int rc = h->b.Build(*pic);
  if (rc) {
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}

void xvc_host_csb_destroy(xvc_csb *b) { delete b; }

const void *xvc_host_csb_array(const xvc_csb *b, int which, int64_t *bytes) {
  int64_t dummy;
  return b ? b->b.Array(which, bytes ? bytes : &dummy) : nullptr;
}

int32_t xvc_host_csb_n_start_dist(const xvc_csb *b) { return b ? b->b.n_start_dist : 0; }
int32_t xvc_host_csb_n_bi_slots(const xvc_csb *b) { return b ? b->b.n_bi_slots : 0; }
int64_t xvc_host_csb_n_edist(const xvc_csb *b) { return b ? b->b.n_edist : 0; }

const xvc_cs_op *xvc_host_csb_program(xvc_csb *b, const xvc_csb_addrs *a, const xvc_csb_intra *in,
                                      int32_t first, int32_t n, int32_t flags, int64_t *n_ops) {
  if (!b || !a || !in || !n_ops) return nullptr;
  return b->b.Program(*a, *in, first, n, flags, n_ops);
}

}  // extern "C"
