"""Per-function table from a KC_HOST_SAMPLE run (see csrc/runtime.cpp, sampler_start).

    KC_HOST_SAMPLE=1 python -m kanter_core_amd.build --force
    gpurun -- 'KC_SAMPLE_OUT=gpurun_out/samples.txt python profiles/host_profile_256.py'
    python profiles/host_samples.py gpurun_out/samples.txt          # here: same image, same library files

Each line of the sample file is "module offset nearest-exported-symbol"; offsets are resolved with llvm-symbolizer
against the module file (static functions included), falling back to the exported symbol.
"""
import collections
import os
import subprocess
import sys

SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def main(path, top=45):
    by_mod = collections.defaultdict(list)
    total = 0
    for line in open(path):
        mod, off, sym = line.split()
        if mod.startswith("/root/repo/") is False and "/kanter_core_amd/" in mod:
            mod = "/root/repo/kanter_core_amd/" + mod.split("/kanter_core_amd/")[-1]
        by_mod[mod].append((int(off, 16), sym))
        total += 1
    funcs = collections.Counter()
    mods = collections.Counter()
    for mod, samples in by_mod.items():
        mods[os.path.basename(mod)] += len(samples)
        names = None
        if os.path.exists(mod) and os.path.exists(SYMBOLIZER):
            inp = "".join("0x%x\n" % o for o, _ in samples)
            r = subprocess.run([SYMBOLIZER, "--obj=" + mod, "--functions=linkage", "--demangle", "--no-inlines", "--output-style=GNU"],
                               input=inp, stdout=subprocess.PIPE, text=True)
            out = r.stdout.splitlines()
            if len(out) == 2 * len(samples):
                names = out[0::2]
        for i, (o, sym) in enumerate(samples):
            name = names[i] if names and names[i] not in ("??", "") else sym
            funcs[(os.path.basename(mod), name[:90])] += 1
    print("%d samples" % total)
    print("-- by module")
    for m, c in mods.most_common(12):
        print("%6.2f%%  %s" % (100.0 * c / total, m))
    print("-- by function")
    for (m, f), c in funcs.most_common(top):
        print("%6.2f%%  %-28s %s" % (100.0 * c / total, m, f))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45)
