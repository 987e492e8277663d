"""Instruction histogram around the MFMA loop of one kernel of the built library (no GPU needed).  usage: loop_ops.py <mangled-name substring>"""
import sys, os, subprocess, collections, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_code_object import device_code_objects, LLVM
tmp = tempfile.mkdtemp()
pat = sys.argv[1]
for o in device_code_objects(os.path.join(ROOT, 'spatialaudiogen_amd', 'libsagen_hip.so'), tmp):
    txt = subprocess.run([LLVM + '/llvm-objdump', '-d', o], capture_output=True, text=True).stdout
    pos = 0
    while True:
        a = txt.find('<_ZN5sagen', pos)
        if a < 0:
            break
        e = txt.find('>:', a)
        name = txt[a + 1:e]
        b = txt.find('\n\n', a)
        pos = b if b > 0 else len(txt)
        if pat not in name:
            continue
        L = txt[a:pos].splitlines()
        idx = [i for i, l in enumerate(L) if 'v_mfma' in l]
        if not idx:
            continue
        lo, hi = max(idx[0] - 150, 0), idx[-1] + 200
        ops = collections.Counter(l.strip().split()[0] for l in L[lo:hi] if l.strip() and not l.strip().endswith(':'))
        print(name[:70], 'lines', len(L), 'mfma', len(idx))
        print('   ', ', '.join('%s %d' % kv for kv in ops.most_common(16)))
