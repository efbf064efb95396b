"""Build gate for the persistent sequence kernels (called by paccmann_polymer_amd/csrc/Makefile).

k_seq_fwd / k_seq_bwd are only correct while every workgroup of a row-block group is resident at the same time
(ppk_seq_persist.hip: seq_capacity sizes a launch from hipOccupancyMaxActiveBlocksPerMultiprocessor).  That query
knows registers and LDS, not the private segment: a refactor that spills heavily, needs a dynamic stack or drops the
kernel below two workgroups per CU must fail the BUILD, not a GPU box.  Reads the compiler's
-Rpass-analysis=kernel-resource-usage remarks of one translation unit and checks every kernel whose name matches.
"""
import re
import sys

MAX_SCRATCH = 512          # bytes / lane (shipping build: 380 forward, 188 backward)
MIN_WAVES_PER_SIMD = 2     # 256-thread workgroups: 2 waves per SIMD = 2 workgroups per CU
MAX_LDS = 80 * 1024        # two workgroups share the CU's 160 KiB


def main(path, pattern, max_lds=MAX_LDS):
    text = open(path, errors='replace').read()
    blocks = re.split(r'remark: Function Name: ', text)[1:]
    bad, seen = [], 0
    for b in blocks:
        name = b.split()[0]
        if not re.search(pattern, name):
            continue
        seen += 1

        def field(key):
            m = re.search(re.escape(key) + r':\s*(\S+)', b)
            return m.group(1) if m else None
        scratch, occ, lds, dyn = int(field('ScratchSize [bytes/lane]')), int(field('Occupancy [waves/SIMD]')), int(field('LDS Size [bytes/block]')), field('Dynamic Stack')
        line = f'{name}: scratch {scratch} B/lane, occupancy {occ} waves/SIMD, LDS {lds} B, dynamic stack {dyn}, VGPR spill {field("VGPRs Spill")}'
        print('  ' + line)
        if scratch > MAX_SCRATCH or occ < MIN_WAVES_PER_SIMD or lds > max_lds or dyn != 'False':
            bad.append(line)
    if seen == 0:
        print(f'check_kernel_resources: no kernel matching {pattern!r} in {path}', file=sys.stderr)
        return 1
    if bad:
        print('check_kernel_resources: FAILED (limits: scratch <= %d B/lane, occupancy >= %d waves/SIMD, LDS <= %d B, static stack)' % (MAX_SCRATCH, MIN_WAVES_PER_SIMD, max_lds), file=sys.stderr)
        for l in bad:
            print('  ' + l, file=sys.stderr)
        return 1
    return 0


if __name__ == '__main__':
    # third argument: LDS limit in bytes (the small-batch weights-stationary kernels run ONE 512-thread workgroup per CU:
    # 2 waves per SIMD with up to the CU's whole 160 KiB)
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else r'k_seq_(fwd|bwd)', int(sys.argv[3]) if len(sys.argv) > 3 else MAX_LDS))
