// spmv.hip -- host side of the row-length-adaptive sparse mat-vec (spmv.h): the row classes of a
// matrix set and the one-time cooking of the coefficients.
#include "spmv.h"

namespace g16 {

namespace {

// col / val in place: unit coefficients are flagged in the column index, every other coefficient c
// (storage form: c * 2^256 mod r) becomes c * 2^266 mod r, canonical, packed in the same 8 words
__global__ void __launch_bounds__(256) k_spmv_cook(uint32_t* col, Fr* val, size_t nnz) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nnz) return;
  const Fr c = val[j];
  if (c == Fr::one()) {
    col[j] |= SPMV_ONE;
    return;
  }
  // from_mont256: c * 2^261; times 2^266 / 2^261: c * 2^266
  const Fr29 k = Fr29::from_mont256(c) * Fr29::from_limbs(Fr29::C::C266);
  Fr o;
  k.pack_internal(o.v);
  val[j] = o;
}

}  // namespace

void spmv_cook(uint32_t* col_dev, Fr* val_dev, size_t nnz, hipStream_t stream) {
  if (!nnz) return;
  G16_LAUNCH(k_spmv_cook, ceil_div(nnz, 256), 256, 0, stream, col_dev, val_dev, nnz);
}

void SpmvPlan::build(const uint32_t* const* rp, int nmat_, uint32_t rows_,
                     const std::function<bool(uint32_t)>& keep) {
  nmat = nmat_;
  rows = rows_;
  std::vector<uint32_t> med, huge, off;
  std::vector<uint32_t> med_len;
  std::vector<SpmvTask> tk;
  for (uint32_t i = 0; i < rows; ++i) {
    uint32_t longest = 0;
    for (int q = 0; q < nmat; ++q) {
      const uint32_t len = rp[q][i + 1] - rp[q][i];
      if (len > longest) longest = len;
    }
    if (longest <= SPMV_SHORT) continue;
    if (keep && !keep(i)) continue;
    if (longest <= SPMV_TASK_TERMS) {
      med.push_back(i);
      uint32_t tot = 0;  // lane iterations of the row's group: what a wave runs is the maximum over its 16 rows
      for (int q = 0; q < nmat; ++q) tot += (rp[q][i + 1] - rp[q][i] + SPMV_G - 1) / SPMV_G;
      med_len.push_back(tot);
      continue;
    }
    huge.push_back(i);
    for (int q = 0; q < nmat; ++q) {
      off.push_back((uint32_t)tk.size());
      for (uint32_t s = rp[q][i]; s < rp[q][i + 1]; s += SPMV_TASK_TERMS) {
        const uint32_t len = rp[q][i + 1] - s < SPMV_TASK_TERMS ? rp[q][i + 1] - s : SPMV_TASK_TERMS;
        tk.push_back(SpmvTask{s, len | ((uint32_t)q << 16)});
      }
    }
  }
  off.push_back((uint32_t)tk.size());
  {
    // rows of equal trip count side by side (counting sort: trip counts are <= nmat * SPMV_LANE_TERMS): the
    // Poseidon chain's 3- to 61-term rows otherwise run every wave at its longest row
    const uint32_t kmax = (uint32_t)nmat * SPMV_LANE_TERMS + 1;
    std::vector<uint32_t> start(kmax + 1, 0), sorted(med.size());
    for (uint32_t l : med_len) ++start[l + 1];
    for (uint32_t k = 0; k < kmax; ++k) start[k + 1] += start[k];
    for (size_t j = 0; j < med.size(); ++j) sorted[start[med_len[j]]++] = med[j];
    med.swap(sorted);
  }
  n_med = (uint32_t)med.size();
  n_huge = (uint32_t)huge.size();
  n_tasks = (uint32_t)tk.size();
  auto up = [](DevBuf<uint32_t>& d, const std::vector<uint32_t>& h) {
    d.alloc(h.size() ? h.size() : 1);
    if (!h.empty()) G16_HIP(hipMemcpy(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  };
  up(med_rows, med);
  up(huge_rows, huge);
  up(task_off, off);
  tasks.alloc(tk.size() ? tk.size() : 1);
  if (!tk.empty()) G16_HIP(hipMemcpy(tasks.p, tk.data(), tk.size() * sizeof(SpmvTask), hipMemcpyHostToDevice));
  partial.alloc((tk.size() ? tk.size() : 1) * (size_t)f29::N);
}

}  // namespace g16
