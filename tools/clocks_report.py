"""profiles/rN_clocks.txt from a bench line: the shader clock the chip held under every matrix-pipe kernel class of the step
(roofline.per_op[*].clock_ghz: sm_clock_stamp before and after each launch on its stream, d(s_memtime) / d(s_memrealtime) x 100 MHz,
median over the compute units both stamps reached), in the step (both queues running) and alone (weight gradients on the main queue).
    python tools/clocks_report.py gpurun_out/r6/bench2.json > profiles/r6_clocks.txt"""
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
rf = j["roofline"]
alone_clk = rf.get("one_queue", {}).get("per_op_clock_ghz", {})
alone_tf = rf.get("one_queue", {}).get("per_op_tflops", {})
print(f"bench line: {j['value']:.1f} samples/s, {j['ms_per_step']:.3f} ms/step, dense layout; bare-MFMA peak on this box "
      f"{rf.get('peak_measured', {}).get('mfma_bf16_tflops', float('nan')):.0f} TFLOP/s")
print(rf.get("clock_how", ""))
print("frac@clock = achieved / (2.5 PFLOP/s x clock / 2.4 GHz): the op's rate against what the clock under it allows\n")
print(f"{'op':46s} {'launch/step':>11s} {'ms/step':>8s} {'TFLOP/s':>8s} {'clock GHz':>9s} {'frac':>6s} {'frac@clock':>10s} | {'alone TFLOP/s':>13s} {'alone GHz':>9s}")
for g in rf["per_op"]:
    ck = g.get("clock_ghz")
    print(f"{g['op'][:46]:46s} {g['launches_per_step']:11.1f} {g['ms_per_step']:8.3f} {g['achieved_tflops']:8.0f} "
          f"{(ck if ck else float('nan')):9.2f} {g['frac']:6.3f} {(g.get('frac_at_clock') or float('nan')):10.3f} | "
          f"{alone_tf.get(g['op'], float('nan')):13.0f} {(alone_clk.get(g['op']) or float('nan')):9.2f}")
for g in rf.get("other_kernels_clock", []):
    ck = g.get("clock_ghz")
    print(f"{g['op'][:46]:46s} {g['launches_per_step']:11.1f} {g['ms_per_step']:8.3f} {'':8s} {(ck if ck else float('nan')):9.2f} {'':6s} {'':10s} | "
          f"{'':13s} {(alone_clk.get(g['op']) or float('nan')):9.2f}")
