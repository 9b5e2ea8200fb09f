# A: the two tap masks leave VCC / s[2:3] for vector registers BEFORE the burst of 24 image gathers (they were read behind it)
i = next(k for k, l in enumerate(K) if "v_cmp_gt_f32_e64 s[2:3], s82, v42" in l and "v_cmp_gt_f32_e32 vcc, s82, v43" in K[k - 1])
j = next(k for k in range(i, len(K)) if "v_cndmask_b32_e64 v21, 1.0, 0, vcc" in K[k])
assert "v_cndmask_b32_e64 v15, 1.0, 0, s[2:3]" in K[j + 1]
mv = K[j:j + 2]
del K[j:j + 2]
K[i + 1:i + 1] = ["\ts_nop 3"] + mv
