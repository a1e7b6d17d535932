"""Generate a straight-line VALU body that looks like the FEM kernels (not like a peak benchmark): NI fp32 FMA / add / mul
instructions over NR live registers with a bounded dependency distance, to measure what a SIMD really sustains per
instruction at 1..8 waves (tools/micro/valu_dag.hip)."""
import random, sys
random.seed(7)
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 48
NI = int(sys.argv[2]) if len(sys.argv) > 2 else 256
DIST = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 0: random operands; k > 0: every instruction reads the result k instructions back
lines = []
last = list(range(NR))
hist = []
for i in range(NI):
    d = random.randrange(NR) if DIST == 0 else (hist[-DIST] if len(hist) >= DIST else random.randrange(NR))
    a, b = random.randrange(NR), random.randrange(NR)
    t = random.randrange(NR)
    if DIST and len(hist) >= DIST:
        a = hist[-DIST]
    op = random.random()
    if op < 0.55:
        lines.append(f"r[{t}] = fmaf(r[{a}], r[{b}], r[{d}]);")
    elif op < 0.8:
        lines.append(f"r[{t}] = r[{a}] - r[{b}];")
    else:
        lines.append(f"r[{t}] = r[{a}] * c{random.randrange(4)};")
    hist.append(t)
print("#define NR %d\n#define NI %d\n#define BODY \\\n" % (NR, NI) + " \\\n".join(lines))
