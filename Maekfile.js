#!/usr/bin/env node
// Build description of the HIP library and of bin/cube in the shape the reference builds itself with:
// one JavaScript file run by node, rules made by CPP / LINK helpers (Maekfile.js:222,309 of the
// reference) plus a HIP rule for the kernels, and named targets on the command line
// (Maekfile.js:38,82-84: `node Maekfile.js bin/cube`).  The runner below is this repository's own and
// small: a rule is (targets, prerequisites, command); a target is rebuilt when it is missing or older
// than a prerequisite; independent rules run in parallel.
//
//   node Maekfile.js                      everything: libibl_hip.so, libibl_png.so, bin/cube, bin/png_roundtrip
//   node Maekfile.js bin/cube             one target and what it needs
//   node Maekfile.js -n [targets]         print the commands, run nothing
//   node Maekfile.js -j4 ...              at most 4 commands at a time (default 8)
//   OUT=<dir> node Maekfile.js ...        objects and products under <dir> instead of vulkan-renderer_amd/
//                                         (prerequisites that already exist in the tree are then used from there)
'use strict';
const fs = require('fs');
const path = require('path');
const { spawn } = require('child_process');

const ROOT = __dirname;
const PKG = 'vulkan-renderer_amd';
const OUT = process.env.OUT ? path.resolve(process.env.OUT) : path.join(ROOT, PKG);
const OBJ = path.join(OUT, 'csrc', 'build');
const HIPCC = process.env.HIPCC || '/opt/rocm/bin/hipcc';
const CXX = process.env.CXX || 'g++';

// ---- runner -------------------------------------------------------------------------------------
const rules = [];           // { targets, deps, cmd, desc }
const byTarget = new Map();
function RULE(targets, deps, cmd, desc) {
    const r = { targets: targets.map((t) => path.resolve(ROOT, t)), deps: deps.map((d) => path.resolve(ROOT, d)), cmd, desc };
    rules.push(r);
    for (const t of r.targets) byTarget.set(t, r);
    return r.targets[0];
}
function mtime(f) { try { return fs.statSync(f).mtimeMs; } catch (e) { return null; } }

async function build(goals, { dry, jobs }) {
    const state = new Map();      // rule -> promise
    let running = 0; const waiting = [];
    const slot = () => new Promise((ok) => { if (running < jobs) { running++; ok(); } else waiting.push(ok); });
    const release = () => { const n = waiting.shift(); if (n) n(); else running--; };
    function need(file) {
        const r = byTarget.get(file);
        if (!r) {
            if (mtime(file) === null) throw new Error('no rule to make ' + path.relative(ROOT, file));
            return Promise.resolve(false);
        }
        if (!state.has(r)) state.set(r, run(r));
        return state.get(r);
    }
    async function run(r) {
        const rebuilt = await Promise.all(r.deps.map(need));
        const newest = Math.max(0, ...r.deps.map((d) => mtime(d) || Infinity));
        const stale = rebuilt.some(Boolean) || r.targets.some((t) => { const m = mtime(t); return m === null || m < newest; });
        if (!stale) return false;
        console.log((dry ? '' : '[maek] ') + (dry ? r.cmd.join(' ') : r.desc));
        if (dry) return true;
        for (const t of r.targets) fs.mkdirSync(path.dirname(t), { recursive: true });
        await slot();
        try {
            await new Promise((ok, fail) => {
                const p = spawn(r.cmd[0], r.cmd.slice(1), { cwd: ROOT, stdio: 'inherit' });
                p.on('error', fail);
                p.on('exit', (code) => (code === 0 ? ok() : fail(new Error(r.desc + ': exit ' + code))));
            });
        } finally { release(); }
        return true;
    }
    await Promise.all(goals.map((g) => need(path.resolve(ROOT, g))));
}

// ---- rule makers (the reference's vocabulary) --------------------------------------------------
const INC = ['-I' + path.join(ROOT, 'include'), '-I' + path.join(ROOT, PKG, 'csrc')];
const HIP_HEADERS = [`${PKG}/csrc/ibl_device.h`, `${PKG}/csrc/ibl_kernels.h`, `${PKG}/csrc/ggx_common.h`, 'include/ibl_hip.h'];

// A kernel source to an object for gfx950.
function HIP(src, extra = []) {
    const obj = path.join(OBJ, path.basename(src, '.hip') + '.o');
    return RULE([obj], [src, ...HIP_HEADERS],
        [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wall',
         '-Wno-unused-result', ...INC, ...extra, '-c', path.join(ROOT, src), '-o', obj], 'HIP ' + src);
}
// A host source to an object.
function CPP(src, deps = [], extra = []) {
    const obj = path.join(OUT, 'host', 'build', path.basename(src, '.cpp') + '.o');
    return RULE([obj], [src, 'include/ibl_hip.h', ...deps],
        [CXX, '-O2', '-std=c++20', '-Wall', '-Wextra', '-fPIC', ...INC, ...extra, '-c', path.join(ROOT, src), '-o', obj], 'CPP ' + src);
}
// Objects to an executable or (with -shared among the flags) a library.
function LINK(objs, out, { linker = CXX, flags = [], libs = [], deps = [] } = {}) {
    const target = path.join(OUT, out);
    return RULE([target], [...objs, ...deps], [linker, ...flags, '-o', target, ...objs, ...libs], 'LINK ' + out);
}

// ---- the targets ---------------------------------------------------------------------------------
const kernels = ['lambertian', 'lambertian_octets', 'ggx', 'ggx_octets', 'ggx_hex', 'ggx_sampled', 'lut_codec', 'png_rows',
                 'sharding', 'mirror_reproject', 'capi'].map((k) => HIP(`${PKG}/csrc/${k}.hip`));
// (the SLP vectoriser turns this kernel's scalar adds into packed adds plus the moves that feed them)
kernels.push(HIP(`${PKG}/csrc/lambertian_split.hip`, ['-fno-slp-vectorize']));
kernels.push(HIP(`${PKG}/csrc/lambertian_hex.hip`, ['-fno-slp-vectorize']));
kernels.push(HIP(`${PKG}/csrc/lambertian_o48.hip`, ['-fno-slp-vectorize']));
const libibl = LINK(kernels, 'libibl_hip.so', { linker: HIPCC, flags: ['--offload-arch=gfx950', '-shared', '-fPIC'] });

const host = `${PKG}/host`;
const png_io = CPP(`${host}/png_io.cpp`, [`${host}/png_io.hpp`]);
const formats = CPP(`${host}/image_formats.cpp`, [`${host}/png_io.hpp`]);
const scene = CPP(`${host}/s72_scene.cpp`, [`${host}/s72_scene.hpp`]);
const cube_o = CPP(`${host}/cube.cpp`, [`${host}/png_io.hpp`, `${host}/s72_scene.hpp`]);
// With OUT set and no library built there, bin/cube links the one in the tree.
const treeLib = path.join(ROOT, PKG, 'libibl_hip.so');
const useTree = process.env.OUT && mtime(treeLib) !== null && process.env.MAEK_LINK_TREE_LIB === '1';
const libDir = useTree ? path.join(ROOT, PKG) : OUT;
// bin/cube: the tool, its image and scene readers, zlib and the HIP library -- none of the renderer
// (the reference links its tool against every object of the viewer, Maekfile.js:78-83).
LINK([cube_o, png_io, formats, scene], 'bin/cube', {
    libs: ['-L' + libDir, '-libl_hip', '-lz', '-pthread', '-Wl,-rpath,' + libDir, '-Wl,-rpath,/opt/rocm/lib'],
    deps: [useTree ? treeLib : libibl],
});
LINK([CPP(`${host}/png_roundtrip.cpp`, [`${host}/png_io.hpp`]), png_io], 'bin/png_roundtrip', { libs: ['-lz', '-pthread'] });
LINK([CPP(`${host}/png_capi.cpp`, [`${host}/png_io.hpp`]), png_io, formats], 'libibl_png.so', { flags: ['-shared', '-fPIC'], libs: ['-lz', '-pthread'] });

// ---- command line ------------------------------------------------------------------------------
const args = process.argv.slice(2);
const opt = { dry: false, jobs: 8 };
const goals = [];
for (const a of args) {
    if (a === '-n') opt.dry = true;
    else if (/^-j\d+$/.test(a)) opt.jobs = Math.max(1, parseInt(a.slice(2), 10));
    else if (a === '-h' || a === '--help') { console.log('usage: node Maekfile.js [-n] [-jN] [targets: libibl_hip.so libibl_png.so bin/cube bin/png_roundtrip]'); process.exit(0); }
    else goals.push(path.join(OUT, a));
}
if (goals.length === 0) goals.push(...['libibl_hip.so', 'libibl_png.so', 'bin/cube', 'bin/png_roundtrip'].map((t) => path.join(OUT, t)));
build(goals, opt).catch((e) => { console.error('[maek] FAILED: ' + e.message); process.exit(1); });
