"""Per-kernel digests of the gfx950 code objects inside libsmolmc_hip.so.

profiles/pmc_constants.json stamps every counter entry with the sha256 of the machine code of the
kernel it was collected on (function bytes + kernel descriptor); bench.py marks an entry
``pmc_stale`` when the kernel of the library it runs has other bytes.  A digest over the source
tree (engine.source_digest, kept as a second stamp) goes stale on a comment and says nothing about
WHICH kernel moved; this one follows the instruction stream.

No external tool: the library's ``.hip_fatbin`` section is a sequence of clang offload bundles
(``__CLANG_OFFLOAD_BUNDLE__``, one per translation unit), each holding an AMDGPU ELF whose symbol
table lists the kernels (STT_FUNC in .text) and their descriptors (``<name>.kd`` in .rodata).
Names are demangled with c++filt when it is on PATH (rocprofv3 reports demangled names)."""

import hashlib
import os
import shutil
import struct
import subprocess

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_CACHE = {}


def _code_objects(blob):
    """The device ELFs of every offload bundle in ``blob`` (bytes of the shared library)."""
    pos = blob.find(_MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", blob, pos + len(_MAGIC))
        q = pos + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if triple.startswith("hip") and "amdgcn" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos = blob.find(_MAGIC, pos + len(_MAGIC))


def _elf_symbols(elf):
    """(name, type, section index, value, size) of the symbol table, and a reader of section bytes."""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return [], None
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    sec = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    # (name, type, flags, addr, offset, size, link, info, addralign, entsize)
    syms = []
    for s in sec:
        if s[1] != 2:  # SHT_SYMTAB
            continue
        stroff = sec[s[6]][4]
        for k in range(s[5] // 24):
            nm, info, _, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, s[4] + 24 * k)
            end = elf.index(b"\0", stroff + nm)
            syms.append((elf[stroff + nm:end].decode(), info & 0xF, shndx, value, size))

    def read(shndx, value, size):
        s = sec[shndx]
        if s[1] == 8:  # SHT_NOBITS
            return b""
        o = s[4] + (value - s[3])
        return elf[o:o + size]

    return syms, read


def _demangle(names):
    exe = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not (exe and os.path.exists(exe)):
        return list(names)
    out = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    dem = out.split("\n")[:len(names)]
    return dem if len(dem) == len(names) else list(names)


def _merge_copies(copies):
    """{mangled name: set of digests of its copies} -> one digest per name: the copy's own when all copies are
    byte-identical, else a digest over all of them (order-free)."""
    return {n: (next(iter(c)) if len(c) == 1 else hashlib.sha256("".join(sorted(c)).encode()).hexdigest())
            for n, c in copies.items()}


def kernel_digests(path=None):
    """{demangled kernel name: sha256 over its machine code and its kernel descriptor} for every
    kernel of the library (default: the library engine.load_library would load)."""
    if path is None:
        from . import engine
        path = os.environ.get("SMOLMC_LIB") or engine.LIB_PATH
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    if key in _CACHE:
        return _CACHE[key]
    with open(path, "rb") as fh:
        blob = fh.read()
    # A kernel defined in a shared header (the static __global__ helpers of smolmc_common.h) is emitted by several
    # translation units under ONE mangled name: every copy is digested, and a name whose copies differ gets the
    # digest of all of them together (sorted) -- it changes when ANY copy changes, so `isa_stale` cannot compare an
    # entry with the wrong translation unit's copy.
    copies = {}
    for elf in _code_objects(blob):
        syms, read = _elf_symbols(elf)
        if read is None:
            continue
        kd = {n[:-3]: (sh, v, sz) for n, t, sh, v, sz in syms if n.endswith(".kd") and 0 < sh < 0xFF00}
        for n, t, sh, v, sz in syms:
            if t != 2 or n not in kd or not (0 < sh < 0xFF00):  # STT_FUNC with a descriptor = a kernel
                continue
            h = hashlib.sha256()
            h.update(read(sh, v, sz))
            h.update(read(*kd[n]))
            copies.setdefault(n, set()).add(h.hexdigest())
    raw = _merge_copies(copies)
    names = sorted(raw)
    out = {d: raw[m] for m, d in zip(names, _demangle(names))}
    _CACHE[key] = out
    return out


def _norm(name):
    """rocprofv3 prints 'void f<...>(LeanParams)' (sometimes cut); c++filt the same: compare without
    blanks so that either spelling of the template arguments matches."""
    return name.replace(" ", "")


def find_kernel(fragment, path=None):
    """(name, digest) of the one kernel whose demangled name contains ``fragment`` (a full name, or a
    name cut short by a summary); None when no kernel or more than one matches."""
    f = _norm(fragment)
    hits = [(n, d) for n, d in kernel_digests(path).items() if f in _norm(n)]
    return hits[0] if len(hits) == 1 else None


def isa_stale(entry, path=None):
    """Is the PMC entry ``entry`` (a dict of profiles/pmc_constants.json) older than the kernel it names?
    Entries stamped with ``kernel_symbol`` / ``isa_sha256`` are compared per kernel; older entries fall
    back to the digest over the source tree."""
    if not entry:
        return False
    if entry.get("isa_sha256") and entry.get("kernel_symbol"):
        hit = find_kernel(entry["kernel_symbol"], path)
        return hit is None or hit[1] != entry["isa_sha256"]
    from . import engine
    return entry.get("csrc_sha256") != engine.source_digest()


if __name__ == "__main__":
    import sys

    for name, dig in sorted(kernel_digests(sys.argv[1] if len(sys.argv) > 1 else None).items()):
        print(dig[:16], name[:160])
