# C: the compare of tap 1's weight sum repeated BEHIND the burst from a copy of the sum kept in v70 (VCC rewritten right in front of its use)
i = next(k for k, l in enumerate(K) if "v_cmp_gt_f32_e64 s[2:3], s82, v42" in l and "v_cmp_gt_f32_e32 vcc, s82, v43" in K[k - 1])
j = next(k for k in range(i, len(K)) if "v_cndmask_b32_e64 v21, 1.0, 0, vcc" in K[k])
K[j:j] = ["\tv_cmp_gt_f32_e32 vcc, s82, v70", "\ts_nop 3"]
K[i - 1:i - 1] = ["\tv_mov_b32_e32 v70, v43"]
