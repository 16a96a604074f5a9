"""Write profiles/b32_error_fit.json -- the per-clip maximum mel error of the SHIPPED batched precision over the 64 real-reference goldens of two
32-clip batches, as bench.py's `batched.mel_error_vs_reference` quotes it -- from the output of the GPU test that measured it:

    python -m pytest tests/test_gpu_headline.py -q -rP -k "batch_of_32 and shipped" > log.txt ; python tools/b32_error_fit.py log.txt

The file is stamped with the hash of the sampler's kernel sources (bench.kernel_sources_sha): bench.py reports the fit as null, with the
reason, once a kernel source has changed since (the numbers used to be a constant in bench.py and could go stale silently)."""
import json, math, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

log = open(sys.argv[1]).read()
vals, prec = {}, None
for m in re.finditer(r"batch of 32 \((random2?) checkpoint, shipped (\S+)\): worst \S+; mel max-abs err per golden clip \[(.*?)\]", log):
    prec = m.group(2)
    for c, e in re.findall(r"'(\d+): ([0-9.e+-]+)'", m.group(3)):
        vals[int(c)] = float(e)
if len(vals) != 64:
    sys.exit("expected the 64 per-clip maxima of the two shares at the shipped precision, found %d" % len(vals))
v = [vals[c] for c in sorted(vals)]
mean = sum(v) / len(v)
sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (len(v) - 1))
beta = sd * math.sqrt(6.0) / math.pi
mu = mean - 0.5772156649 * beta
out = {"precision": prec, "clips": len(v), "worst": max(v), "best": min(v), "gumbel_mu": mu, "gumbel_beta": beta,
       "csrc_sha16": bench.kernel_sources_sha(),
       "source": "tests/test_gpu_headline.py::test_batch_of_32_full_chain_every_clip_with_a_golden[random|random2-shipped] (%s)" % os.path.basename(sys.argv[1])}
path = os.path.join(ROOT, "profiles", "b32_error_fit.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(path, out)
