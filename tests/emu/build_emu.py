"""Build tests/emu/_build/librdmi_emu.so: the UNMODIFIED csrc sources compiled by g++ against the CPU
execution-model emulator (tests/emu/include/hip/hip_runtime.h).  Test infrastructure only."""
import fcntl
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'optimized-diffusion-model_amd', 'csrc')
OUT = os.path.join(HERE, '_build', 'librdmi_emu.so')


def build(force=False, sanitize=False):
    out = OUT.replace('.so', '_asan.so') if sanitize else OUT
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(HERE, 'emu.cpp'),
            os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'), os.path.join(ROOT, 'include', 'rdmi.h')]
    def fresh():
        return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs)
    if not force and fresh():
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # parallel test workers (pytest -n) reach this together: one builds (aside, then rename), the others wait on the lock
    with open(out + '.lock', 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if force or not fresh():
            tmp = out + f'.tmp{os.getpid()}'
            cmd = ['g++', '-std=c++17', '-O2', '-g', '-fPIC', '-shared', '-I' + os.path.join(HERE, 'include'), '-x', 'c++',
                   os.path.join(CSRC, 'rdmi.hip'), os.path.join(HERE, 'emu.cpp'), '-o', tmp, '-lpthread']
            if sanitize:
                cmd[3:3] = ['-fsanitize=address', '-fno-omit-frame-pointer']
            subprocess.run(cmd, check=True)
            os.replace(tmp, out)
    return out


if __name__ == '__main__':
    print(build(force=True))
