"""register / scratch report from a hipcc -S listing: python tools/kernel_regs.py file.s [substring]"""
import re
import sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    def get(key):
        r = re.search(r"\.amdhsa_%s (\S+)" % key, body)
        return r.group(1) if r else "?"
    print("%-100s vgpr=%-4s accum_off=%-4s scratch=%-5s" % (name.replace("_ZN2lg10sgemm_mfmaI", "sgemm<").replace("EEvNS_8GemmArgsE", ">"),
          get("next_free_vgpr"), get("accum_offset"), get("private_segment_fixed_size")))
