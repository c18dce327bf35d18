"""Rewrite the round-6 figures of DESIGN.md section 3.2b / 4 / 5 from the files under profiles/ (after tools/copy_r6_profiles.sh)."""
import json
import re

def last(f):
    return json.loads(open('profiles/%s' % f).read().strip().splitlines()[-1])

def ms(f):
    return last('r06_bench_%s.json' % f)['ms_per_step']

s = open('DESIGN.md').read()
d = last('r06_bench.json')

def sub(pattern, new, count=1):
    global s
    s2, n = re.subn(pattern, new, s, count=count)
    assert n >= 1, pattern
    s = s2

num = r"[0-9]+\.[0-9]+"
sub(r"\*\*%s M graph-instances/s, %s ms/step\*\*" % (num, num), "**%.2f M graph-instances/s, %.4f ms/step**" % (d['value'] / 1e6, d['ms_per_step']))
sub(r"`fast_path` \(complement\) %s ms" % num, "`fast_path` (complement) %.4f ms" % d['fast_path']['ms_per_step'])
sub(r"7\.27 GFLOP in [0-9]+ µs \(HIP events, eager\) = %s" % num, "7.27 GFLOP in %.0f µs (HIP events, eager) = %.3f" % (d['roofline']['avg_launch_us'], d['roofline']['frac']))
sub(r"0\.1920 → \*\*%s\*\*" % num, "0.1920 → **%.4f**" % ms('b2048'))
sub(r"0\.1520 → \*\*%s\*\*" % num, "0.1520 → **%.4f**" % ms('b1024'))
sub(r"0\.1206 → \*\*%s\*\*" % num, "0.1206 → **%.4f**" % ms('b512'))
sub(r"\| 512 \| %s → \*\*%s\*\* \|" % (num, num), "| 512 | %.4f → **%.4f** |" % (ms('b512_dense0_in_mlp'), ms('b512')))
sub(r"\| 1024 \| %s → \*\*%s\*\* \|" % (num, num), "| 1024 | %.4f → **%.4f** |" % (ms('b1024_dense0_in_mlp'), ms('b1024')))
sub(r"\| 2048 \| %s → \*\*%s\*\* \|" % (num, num), "| 2048 | %.4f → **%.4f** |" % (ms('b2048_dense0_in_mlp'), ms('b2048')))
sub(r"\| 4096 \| %s → %s \(not chosen\)" % (num, num), "| 4096 | %.4f → %.4f (not chosen)" % (ms('b4096_dense0_in_mlp') if False else d['ms_per_step'], ms('b4096_dense0_as_roles')))
sub(r"Dense-0 as roles at batch 4096 \(%s against %s\)" % (num, num), "Dense-0 as roles at batch 4096 (%.4f against %.4f)" % (ms('b4096_dense0_as_roles'), d['ms_per_step']))
sub(r"0\.2586 → %s \| 5" % num, "0.2586 → %.4f | 5" % d['ms_per_step'])
sub(r"\| 1 \| 0\.2586 → %s ms \|" % num, "| 1 | 0.2586 → %.4f ms |" % d['ms_per_step'])
sub(r"\| 2 \| 0\.192 → %s \|" % num, "| 2 | 0.192 → %.4f |" % ms('b2048'))
sub(r"\| 4 \| 0\.152 → %s \|" % num, "| 4 | 0.152 → %.4f |" % ms('b1024'))
sub(r"\| 8 \| 0\.121 → %s \|" % num, "| 8 | 0.121 → %.4f |" % ms('b512'))
sub(r"%s ms \(305 k graphs/s\) / %s ms" % (num, num), "%.2f ms (305 k graphs/s) / %.3f ms" % (ms('cfg4'), ms('cfg5')))
sub(r"shared weights %s ms; batch 65,536 %s ms" % (num, num), "shared weights %.4f ms; batch 65,536 %.2f ms" % (ms('shared'), ms('b65536')))
l1 = [ms('cfg2loop_env1_run%d' % i) for i in (1, 2, 3)]
sub(r"\*\*[0-9.]+-9\.7 → [0-9.]+-[0-9.]+ ms per train step\*\*", "**%.1f-9.7 → %.1f-%.1f ms per train step**" % (min(9.1, ms('cfg2loop_env1_per_transition')), min(l1), max(l1)))
sub(r"with a B = 1 predict per greedy transition [0-9.]+-3\.6 ms", "with a B = 1 predict per greedy transition %.1f-3.6 ms" % ms('cfg2loop_env1_b1_predicts'))
thr = open('profiles/r06_loop_env1_threads.txt').read().strip().splitlines()
sub(r"by team size \(`profiles/r06_loop_env1_threads\.txt`\): [^\n]*?\.\n", "by team size (`profiles/r06_loop_env1_threads.txt`): " + "; ".join(
    "%s threads %.2f ms" % (l.split(':')[0].split('=')[1], float(l.split(':')[1].split()[0])) for l in thr) + ".\n")
sub(r"\*\*[0-9.]+ / [0-9.]+ / [0-9.]+\*\* ms per train step \(0\.84-1\.13 across the day's boxes; round 5: 0\.97-1\.12\); 4000 train steps [0-9.]+\.",
    "**%s** ms per train step (0.84-1.13 across the day's boxes; round 5: 0.97-1.12); 4000 train steps %.2f." % (
        " / ".join("%.2f" % ms('cfg2loop_envs50_run%d' % i) for i in (1, 2, 3)), ms('cfg2loop_envs50_4000steps')))
open('DESIGN.md', 'w').write(s)
print("headline %.4f ms, shares %.4f / %.4f / %.4f, cfg4 %.2f, cfg5 %.3f" % (d['ms_per_step'], ms('b512'), ms('b1024'), ms('b2048'), ms('cfg4'), ms('cfg5')))
