"""Register / LDS footprint of the built kernels (code-object notes; no GPU needed).  usage: kernel_regs.py [name substring]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_code_object import device_code_objects, LLVM
pat = sys.argv[1] if len(sys.argv) > 1 else ''
tmp = tempfile.mkdtemp()
for o in device_code_objects(os.path.join(ROOT, 'spatialaudiogen_amd', 'libsagen_hip.so'), tmp):
    notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', o], capture_output=True, text=True).stdout
    for blk in notes.split('- .agpr_count:')[1:]:
        name = re.search(r'\.name:\s*(\S+)', blk)
        if not name or pat not in name.group(1):
            continue
        g = lambda k: re.search(r'\.%s:\s*(\d+)' % k, blk)
        dem = name.group(1).replace('_ZN5sagen', '')
        print('%-60s vgpr %s agpr %s sgpr %s lds %s scratch %s spill %s' % (dem[:60], g('vgpr_count').group(1), blk.split()[0], g('sgpr_count').group(1),
              g('group_segment_fixed_size').group(1), g('private_segment_fixed_size').group(1), (g('vgpr_spill_count') or re.match('(0)', '0')).group(1)))
