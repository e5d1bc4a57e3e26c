// Probe: LDS throughput on gfx950 of the operations the weight-gradient kernel lives on, per CU, in shader cycles per wave-instruction:
// ds_read_b64_tr_b16 with the kernel's address pattern (1088-byte fragment stride, rotated image), ds_read_b64, ds_read_b128,
// ds_write_b128, ds_write_b64 -- with 4, 8 and 16 waves per workgroup (one workgroup per CU).  hipcc --offload-arch=gfx950 -O3
// tools/probe_lds.hip -o /tmp/probe_lds && /tmp/probe_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int OP>
__global__ void __launch_bounds__(1024) k(long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, rh = (lane >> 4) & 1, m = (lane >> 2) & 3, q = lane & 3;
  const int point = 8 * hh + m;
  uint32_t tr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + rh * 1088 + ((q >> 1) ? 512 + ((point + 8) & 31) * 16 : point * 16) + (q & 1) * 8 + (wave & 3) * 2176;
  uint32_t lin16 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 16 + (wave & 7) * 1088;
  uint32_t lin8 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + lane * 8 + (wave & 7) * 1088;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {
      asm volatile(
          "ds_read_b64_tr_b16 v[10:11], %0 offset:0\n ds_read_b64_tr_b16 v[12:13], %0 offset:64\n ds_read_b64_tr_b16 v[14:15], %0 offset:2176\n ds_read_b64_tr_b16 v[16:17], %0 offset:2240\n"
          "ds_read_b64_tr_b16 v[18:19], %0 offset:4352\n ds_read_b64_tr_b16 v[20:21], %0 offset:4416\n ds_read_b64_tr_b16 v[22:23], %0 offset:6528\n ds_read_b64_tr_b16 v[24:25], %0 offset:6592\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr) : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","memory");
    } else if constexpr (OP == 1) {
      asm volatile(
          "ds_read_b64 v[10:11], %0 offset:0\n ds_read_b64 v[12:13], %0 offset:1088\n ds_read_b64 v[14:15], %0 offset:2176\n ds_read_b64 v[16:17], %0 offset:3264\n"
          "ds_read_b64 v[18:19], %0 offset:4352\n ds_read_b64 v[20:21], %0 offset:5440\n ds_read_b64 v[22:23], %0 offset:6528\n ds_read_b64 v[24:25], %0 offset:7616\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(lin8) : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","memory");
    } else if constexpr (OP == 2) {
      asm volatile(
          "ds_read_b128 v[10:13], %0 offset:0\n ds_read_b128 v[14:17], %0 offset:1088\n ds_read_b128 v[18:21], %0 offset:2176\n ds_read_b128 v[22:25], %0 offset:3264\n"
          "ds_read_b128 v[26:29], %0 offset:4352\n ds_read_b128 v[30:33], %0 offset:5440\n ds_read_b128 v[34:37], %0 offset:6528\n ds_read_b128 v[38:41], %0 offset:7616\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(lin16) : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","memory");
    } else if constexpr (OP == 3) {
      asm volatile(
          "ds_write_b128 %0, v[10:13] offset:0\n ds_write_b128 %0, v[10:13] offset:1088\n ds_write_b128 %0, v[10:13] offset:2176\n ds_write_b128 %0, v[10:13] offset:3264\n"
          "ds_write_b128 %0, v[10:13] offset:4352\n ds_write_b128 %0, v[10:13] offset:5440\n ds_write_b128 %0, v[10:13] offset:6528\n ds_write_b128 %0, v[10:13] offset:7616\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(lin16) : "v10","v11","v12","v13","memory");
    } else if constexpr (OP == 4) {
      asm volatile(
          "ds_write_b64 %0, v[10:11] offset:0\n ds_write_b64 %0, v[10:11] offset:1088\n ds_write_b64 %0, v[10:11] offset:2176\n ds_write_b64 %0, v[10:11] offset:3264\n"
          "ds_write_b64 %0, v[10:11] offset:4352\n ds_write_b64 %0, v[10:11] offset:5440\n ds_write_b64 %0, v[10:11] offset:6528\n ds_write_b64 %0, v[10:11] offset:7616\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(lin8) : "v10","v11","memory");
    } else if constexpr (OP == 5) {  // the kernel's mix: 4 transposed reads + 1 write
      asm volatile(
          "ds_read_b64_tr_b16 v[10:11], %0 offset:0\n ds_read_b64_tr_b16 v[12:13], %0 offset:64\n ds_read_b64_tr_b16 v[14:15], %0 offset:2176\n ds_read_b64_tr_b16 v[16:17], %0 offset:2240\n"
          "ds_write_b128 %1, v[26:29] offset:40000\n"
          "ds_read_b64_tr_b16 v[18:19], %0 offset:4352\n ds_read_b64_tr_b16 v[20:21], %0 offset:4416\n ds_read_b64_tr_b16 v[22:23], %0 offset:6528\n ds_read_b64_tr_b16 v[24:25], %0 offset:6592\n"
          "ds_write_b128 %1, v[26:29] offset:41088\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","memory");
    }
    else if constexpr (OP == 6) {
      asm volatile(
          "ds_read_b64_tr_b16 v[10:11], %0 offset:0\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_read_b64_tr_b16 v[12:13], %0 offset:64\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_read_b64_tr_b16 v[14:15], %0 offset:2176\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_read_b64_tr_b16 v[16:17], %0 offset:2240\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_read_b64_tr_b16 v[18:19], %0 offset:4352\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_read_b64_tr_b16 v[20:21], %0 offset:4416\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_read_b64_tr_b16 v[22:23], %0 offset:6528\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_read_b64_tr_b16 v[24:25], %0 offset:6592\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 7) {
      asm volatile(
          "ds_read_b64_tr_b16 v[10:11], %0 offset:0\n"
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "ds_read_b64_tr_b16 v[12:13], %0 offset:64\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "ds_read_b64_tr_b16 v[14:15], %0 offset:2176\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "ds_read_b64_tr_b16 v[16:17], %0 offset:2240\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "ds_read_b64_tr_b16 v[18:19], %0 offset:4352\n"
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "ds_read_b64_tr_b16 v[20:21], %0 offset:4416\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "ds_read_b64_tr_b16 v[22:23], %0 offset:6528\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "ds_read_b64_tr_b16 v[24:25], %0 offset:6592\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 8) {
      asm volatile(
          "ds_write_b128 %1, v[26:29] offset:40000\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_write_b128 %1, v[26:29] offset:41088\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_write_b128 %1, v[26:29] offset:42176\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_write_b128 %1, v[26:29] offset:43264\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_write_b128 %1, v[26:29] offset:44352\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_write_b128 %1, v[26:29] offset:45440\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_write_b128 %1, v[26:29] offset:46528\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "ds_write_b128 %1, v[26:29] offset:47616\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 9) {
      asm volatile(
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 10) {
      asm volatile(
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "ds_read_b64_tr_b16 v[10:11], %0 offset:0\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "ds_read_b64_tr_b16 v[12:13], %0 offset:64\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "ds_read_b64_tr_b16 v[14:15], %0 offset:2176\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "ds_read_b64_tr_b16 v[16:17], %0 offset:2240\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "ds_write_b128 %1, v[26:29] offset:43264\n"
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "ds_read_b64_tr_b16 v[18:19], %0 offset:4352\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "ds_read_b64_tr_b16 v[20:21], %0 offset:4416\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "ds_read_b64_tr_b16 v[22:23], %0 offset:6528\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "ds_read_b64_tr_b16 v[24:25], %0 offset:6592\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "ds_write_b128 %1, v[26:29] offset:47616\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 11) {
      asm volatile(
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "v_fma_f32 v30, v112, v113, v30\n"
          "v_fma_f32 v31, v112, v113, v31\n"
          "v_fma_f32 v32, v112, v113, v32\n"
          "v_fma_f32 v33, v112, v113, v33\n"
          "v_fma_f32 v34, v112, v113, v34\n"
          "v_fma_f32 v35, v112, v113, v35\n"
          "v_fma_f32 v36, v112, v113, v36\n"
          "v_fma_f32 v37, v112, v113, v37\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 12) {
      asm volatile(
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 13) {
      asm volatile(
          "ds_write_b128 %1, v[26:29] offset:40000\n"
          "ds_write_b128 %1, v[26:29] offset:41088\n"
          "ds_write_b128 %1, v[26:29] offset:42176\n"
          "ds_write_b128 %1, v[26:29] offset:43264\n"
          "ds_write_b128 %1, v[26:29] offset:44352\n"
          "ds_write_b128 %1, v[26:29] offset:45440\n"
          "ds_write_b128 %1, v[26:29] offset:46528\n"
          "ds_write_b128 %1, v[26:29] offset:47616\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
    else if constexpr (OP == 14) {
      asm volatile(
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "ds_read_b64_tr_b16 v[10:11], %0 offset:0\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "ds_read_b64_tr_b16 v[12:13], %0 offset:64\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "ds_read_b64_tr_b16 v[14:15], %0 offset:2176\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "ds_read_b64_tr_b16 v[16:17], %0 offset:2240\n"
          "ds_write_b128 %1, v[26:29] offset:43264\n"
          "v_mfma_f32_32x32x16_bf16 v[40:55], v[104:107], v[108:111], v[40:55]\n"
          "ds_read_b64_tr_b16 v[18:19], %0 offset:4352\n"
          "v_mfma_f32_32x32x16_bf16 v[56:71], v[104:107], v[108:111], v[56:71]\n"
          "ds_read_b64_tr_b16 v[20:21], %0 offset:4416\n"
          "v_mfma_f32_32x32x16_bf16 v[72:87], v[104:107], v[108:111], v[72:87]\n"
          "ds_read_b64_tr_b16 v[22:23], %0 offset:6528\n"
          "v_mfma_f32_32x32x16_bf16 v[88:103], v[104:107], v[108:111], v[88:103]\n"
          "ds_read_b64_tr_b16 v[24:25], %0 offset:6592\n"
          "ds_write_b128 %1, v[26:29] offset:47616\n"
          "s_waitcnt lgkmcnt(0)\n" ::"v"(tr), "v"(lin16) : "memory", "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119");
    }
  }
  __syncthreads();  // every wave of the workgroup is done: the time is the slowest wave's
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int per_iter) {
  long long* d;
  hipMalloc(&d, 8 * 256);
  const int iters = 2000;
  for (int waves : {4, 8, 16}) {
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * waves), 65536, 0, d, iters);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 256; ++i) cyc += h[i];
    cyc /= 256;
    printf("%-28s %2d waves/CU: %6.2f cycles per wave-instruction, %6.2f per CU-instruction\n", name, waves, cyc / iters / per_iter, cyc / iters / per_iter / waves);
  }
  hipFree(d);
}

int main() {
  run<0>("ds_read_b64_tr_b16", 8);
  run<1>("ds_read_b64", 8);
  run<2>("ds_read_b128", 8);
  run<3>("ds_write_b128", 8);
  run<4>("ds_write_b64", 8);
  run<5>("mix 8 tr reads + 2 writes", 10);
  run<6>("8 tr reads + 32 VALU", 8);
  run<7>("8 tr reads + 8 MFMA", 8);
  run<8>("8 writes + 96 VALU", 8);
  run<9>("8 MFMA + 48 VALU", 8);
  run<10>("8 MFMA + 8 tr + 2 wr + 40 VALU", 8);
  run<11>("32 VALU only", 8);
  run<12>("8 MFMA only", 8);
  run<13>("8 writes only (b128)", 8);
  run<14>("8 MFMA + 8 tr + 2 wr", 8);
  return 0;
}
