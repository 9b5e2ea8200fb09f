# B: VCC copied to s[90:91] in front of the burst and the mask taken from the copy behind it (is it VCC that changes, or any SGPR pair?)
i = next(k for k, l in enumerate(K) if "v_cmp_gt_f32_e64 s[2:3], s82, v42" in l and "v_cmp_gt_f32_e32 vcc, s82, v43" in K[k - 1])
j = next(k for k in range(i, len(K)) if "v_cndmask_b32_e64 v21, 1.0, 0, vcc" in K[k])
K[j] = "\tv_cndmask_b32_e64 v21, 1.0, 0, s[90:91]"
K[i + 1:i + 1] = ["\ts_nop 3", "\ts_mov_b64 s[90:91], vcc", "\ts_nop 3"]
