"""CPU helper: disassemble the built objects' gfx950 code and list, per kernel, the two patterns that cost exposed round trips
in this repo's kernels:
  (1) a vector-memory load followed within a few instructions by `s_waitcnt vmcnt(0)` (a load parked under a lane-divergent
      branch, or a dependent address) -- count and first source offsets;
  (2) scalar loads (`s_load_*`) that appear BEHIND the kernel's first barrier (kernel arguments / scalar memory re-read in
      front of a phase).
usage: python tools/isa_audit.py [kernel-name regex]"""
import atexit, os, re, shutil, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LL = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
tmp = tempfile.mkdtemp()
atexit.register(shutil.rmtree, tmp, ignore_errors=True)
for o in sorted(os.listdir(os.path.join(R, "dgcnn_amd/csrc"))):
    if not o.endswith(".o"): continue
    b = o[:-2]
    fat, co = os.path.join(tmp, b + ".fat"), os.path.join(tmp, b + ".co")
    if subprocess.run([f"{LL}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", os.path.join(R, "dgcnn_amd/csrc", o)],
                      capture_output=True).returncode: continue
    if subprocess.run([f"{LL}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                       f"--input={fat}", f"--output={co}", "--unbundle"], capture_output=True).returncode: continue
    dis = subprocess.run([f"{LL}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout.splitlines()
    name, body = None, []
    def report():
        if not name or not pat.search(name): return
        ins = [l.split("//")[0].strip() for l in body if l.startswith("\t")]
        tight, after_bar, seen_bar = 0, 0, False
        for i, t in enumerate(ins):
            if t.startswith("s_barrier"): seen_bar = True
            if t.startswith("s_load_") and seen_bar: after_bar += 1
            if t.startswith("global_load") or t.startswith("flat_load") or t.startswith("buffer_load"):
                for j in range(i + 1, min(i + 4, len(ins))):
                    if ins[j].startswith("s_waitcnt vmcnt(0)"): tight += 1; break
                    if ins[j].startswith(("global_load", "flat_load", "buffer_load")): break
        short = re.sub(r"^_Z\d+", "", name)[:70]
        print(f"{short:70s} instr {len(ins):6d}  load->vmcnt(0) {tight:3d}  s_load behind a barrier {after_bar:3d}")
    for l in dis:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", l)
        if m:
            report(); name, body = m.group(1), []
        else: body.append(l)
    report()
