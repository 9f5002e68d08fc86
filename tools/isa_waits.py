#!/usr/bin/env python3
"""Read a hipcc -S listing: per kernel, what the wave waits for right in front of its MFMAs.  For every v_mfma the nearest preceding
s_waitcnt (with no other MFMA in between) is classified: lgkmcnt(N) = N DS reads still in flight when the MFMA may issue (0 = the MFMA waits
out the round trip of the read issued just before it), vmcnt(N) likewise for vector memory; scratch accesses and vmcnt(0) between the first and
the last MFMA are counted.  Round 5: this is how the one-read-ahead first GEMM of the bf16 fused FFN was found.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -S --cuda-device-only csrc/x3_strip.hip -o /tmp/x.s && python tools/isa_waits.py /tmp/x.s [filter]"""
import collections, re, subprocess, sys

src = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
names = re.findall(r'^(_Z\w+):\s*; @', src, flags=re.M)
def demangle(n):
    try:
        return subprocess.run(['/usr/bin/c++filt', n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
for n in names:
    i = src.index(n + ':'); j = src.index('.Lfunc_end', i)
    body = [l.strip() for l in src[i:j].split('\n')]
    fm = [k for k, l in enumerate(body) if l.startswith('v_mfma')]
    if len(fm) < 8:
        continue
    dn = re.sub(r'\(anonymous namespace\)::', '', demangle(n)); dn = re.sub(r'\(.*$', '', dn).replace('void ', '')
    if flt and flt not in dn:
        continue
    lg, vm = collections.Counter(), collections.Counter()
    for a, k in enumerate(fm):
        lo = fm[a - 1] if a else max(0, k - 40)
        w = [l for l in body[lo + 1:k] if l.startswith('s_waitcnt')]
        if not w:
            lg['none'] += 1
            continue
        m = re.search(r'lgkmcnt\((\d+)\)', ' '.join(w)); v = re.search(r'vmcnt\((\d+)\)', ' '.join(w))
        lg[int(m.group(1)) if m else 'none'] += 1
        if v: vm[int(v.group(1))] += 1
    inner = body[fm[0]:fm[-1]]
    scr = sum(1 for l in inner if l.startswith('scratch_'))
    vm0 = sum(1 for l in inner if l.startswith('s_waitcnt') and 'vmcnt(0)' in l)
    print('%-64s mfma %4d | lgkmcnt before an MFMA: %s | vmcnt before an MFMA: %s | between the MFMAs: scratch %d, vmcnt(0) %d' % (
        dn[:64], len(fm), dict(sorted(lg.items(), key=lambda kv: str(kv[0]))), dict(sorted(vm.items())), scr, vm0))
