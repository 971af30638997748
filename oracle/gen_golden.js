#!/usr/bin/env node
// TEST INFRASTRUCTURE (oracle side). Not part of the product path.
//
// Golden-vector generator: executes the reference component's OWN JavaScript
// (/root/reference/index.js, read by absolute path, never copied) under node's
// `vm` with a stub AFRAME and the build-authored THREE stand-in, drives
//   * createWorker / sortSplats      (index.js:488-599)
//   * pushDataBuffer                 (index.js:328-437)
//   * processPlyBuffer               (index.js:600-745)
//   * tick / getModelViewMatrix / getProjectionMatrix / onBeforeRender uniforms
//                                    (index.js:184-195, 438-487)
// on small seeded inputs and writes inputs + outputs as raw little-endian
// arrays under tests/golden/ (one .bin per case + manifest.json).
//
// Runs only in the build container (needs /root/reference).  On the GPU box the
// committed fixtures are used and this script is never executed.
//
//   node oracle/gen_golden.js            # regenerate tests/golden/*
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const THREE = require('./three_standin.js');

const REF = '/root/reference/index.js';
const OUT = path.join(__dirname, '..', 'tests', 'golden');
if (!fs.existsSync(REF)) { console.error('reference not present; nothing to do'); process.exit(0); }
fs.mkdirSync(OUT, { recursive: true });

// ---------------------------------------------------------------- load reference
let def;
const quiet = { log() {}, error() {}, time() {}, timeEnd() {} };
const ctx = {
  AFRAME: { registerComponent: (n, d) => { def = d; } },
  THREE: Object.assign({}, THREE),
  console: quiet, TextDecoder, Math, parseInt, Float32Array, Uint8Array, Uint32Array, Int16Array,
  Int32Array, Uint8ClampedArray, DataView, ArrayBuffer, Proxy, Error, Promise, setTimeout, Infinity,
};
vm.createContext(ctx);
vm.runInContext(fs.readFileSync(REF, 'utf8'), ctx, { filename: REF });

// ---------------------------------------------------------------- helpers
function rng(seed) {          // mulberry32
  let a = seed >>> 0;
  const u = () => { a = (a + 0x6D2B79F5) >>> 0; let t = a; t = Math.imul(t ^ (t >>> 15), t | 1);
    t ^= t + Math.imul(t ^ (t >>> 7), t | 61); return ((t ^ (t >>> 14)) >>> 0) / 4294967296; };
  const n = () => { const r = Math.sqrt(-2 * Math.log(1 - u())); return r * Math.cos(2 * Math.PI * u()); };
  return { u, n };
}
const DT = new Map([[Float32Array, 'f4'], [Float64Array, 'f8'], [Uint32Array, 'u4'], [Int32Array, 'i4'],
  [Uint8Array, 'u1'], [Int16Array, 'i2'], [Uint16Array, 'u2']]);
const manifest = {};
function emit(name, kind, arrays, meta) {
  const chunks = []; let off = 0; const desc = {};
  for (const k of Object.keys(arrays)) {
    const a = arrays[k]; const dt = DT.get(a.constructor);
    if (!dt) throw new Error('dtype? ' + k);
    const b = Buffer.from(a.buffer, a.byteOffset, a.byteLength);
    desc[k] = { dtype: dt, count: a.length, offset: off };
    chunks.push(b); off += b.length;
    const pad = (8 - (off % 8)) % 8; if (pad) { chunks.push(Buffer.alloc(pad)); off += pad; }
  }
  fs.writeFileSync(path.join(OUT, name + '.bin'), Buffer.concat(chunks));
  manifest[name] = { kind, arrays: desc, meta: meta || {} };
}

// ---------------------------------------------------------------- SORT cases
function runWorker(pushes, view, cutout) {
  let out;
  const self = { postMessage: (m) => { out = m; } };
  def.createWorker(self);
  self.onmessage({ data: { method: 'clear' } });
  for (const m of pushes) self.onmessage({ data: { method: 'push', matrices: m.slice().buffer } });
  self.onmessage({ data: { method: 'sort', view: new Float32Array(view).buffer,
    cutout: cutout ? new Float32Array(cutout) : undefined } });
  return new Uint32Array(out.sortedIndexes);
}
function expand(rows4) {   // worker rows: only floats 12..15 of each 16 are read (index.js:520-548)
  const n = rows4.length / 4; const m = new Float32Array(n * 16);
  for (let i = 0; i < n; i++) for (let k = 0; k < 4; k++) m[i * 16 + 12 + k] = rows4[i * 4 + k];
  return m;
}
function sortCase(name, rows4list, view, cutout, note) {
  const pushes = rows4list.map(expand);
  const idx = runWorker(pushes, view, cutout);
  const all = new Float32Array(rows4list.reduce((a, r) => a + r.length, 0)); let o = 0;
  for (const r of rows4list) { all.set(r, o); o += r.length; }
  const arrays = { rows4: all, view: new Float32Array(view), sorted: idx };
  if (cutout) arrays.cutout = new Float32Array(cutout);
  emit(name, 'sort', arrays, { note, pushes: rows4list.map((r) => r.length / 4), has_cutout: !!cutout });
}
function cloud(seed, n, opt) {
  const r = rng(seed); const rows = new Float32Array(n * 4); opt = opt || {};
  const sp = opt.spread || [2.5, 1.0, 2.5]; const c = opt.center || [0, 0, 0];
  for (let i = 0; i < n; i++) {
    rows[i * 4 + 0] = c[0] + sp[0] * r.n(); rows[i * 4 + 1] = c[1] + sp[1] * r.n(); rows[i * 4 + 2] = c[2] + sp[2] * r.n();
    const s = Math.exp(-4.2 + 0.9 * r.n()); const a = Math.floor(256 * r.u());
    rows[i * 4 + 3] = (opt.tiny && r.u() < 0.5 ? s * 1e-4 * r.u() : s * a / 255.0) * (opt.sizeScale || 1);
  }
  return rows;
}
const V_FRONT = [0.0, 0.0, 1.0, -6.0];   // depth = z - 6 : everything with z<6 is in front
sortCase('sort_n1', [new Float32Array([0.5, -0.25, 1.0, 0.01])], V_FRONT, null, 'single splat in front');
sortCase('sort_equal3', [new Float32Array([1, 0, 2, .1, -1, 3, 2, .1, 0, -2, 2, .1])], V_FRONT, null,
  'equal depths: depthInv=Infinity, NaN|0 -> identity order');
sortCase('sort_n257', [cloud(11, 257)], [0.12, -0.35, 0.93, -7.5], null, 'random, odd N');
sortCase('sort_n4096', [cloud(12, 4096)], [0.6, 0.1, 0.79, -5.0], null, 'random 4096, part behind camera');
sortCase('sort_n4096_cutout', [cloud(13, 4096)], [-0.3, 0.2, 0.93, -6.0],
  // column-major object->cutout-box (unit box) matrix: scale + shear + offset
  [0.31, 0.02, 0, 0, 0.01, 0.45, 0.03, 0, -0.02, 0, 0.27, 0, 0.1, -0.05, 0.2, 1], 'box cutout (index.js:526-545)');
sortCase('sort_n4096_cutout_persp', [cloud(14, 4096)], [0.0, 0.0, 1.0, -9.0],
  [0.3, 0, 0, 0.01, 0, 0.4, 0, -0.02, 0, 0, 0.3, 0.015, 0, 0, 0, 1], 'cutout with non-trivial w divide');
sortCase('sort_all_behind', [cloud(15, 257)], [0, 0, 1, 40.0], null, 'all depth>=0 -> empty result');
sortCase('sort_narrow_far', [cloud(16, 4096, { spread: [1e-3, 1e-3, 1e-3], center: [3, 2, -500], sizeScale: 40 })],
  [0.01, 0.02, 0.9997, -250.0], null, 'narrow depth range far away: f32 rounding of stored depth matters');
sortCase('sort_oob_bucket', [cloud(21, 2048, { spread: [1e-5, 1e-5, 1e-5], center: [3, 2, -500], sizeScale: 40 })],
  [0.01, 0.02, 0.9997, -250.0], null,
  'pathological: f32 rounding of stored depth >> depth range -> buckets <0 or >65535 are silently dropped by the typed-array writes (index.js:561-567); output keeps length V with trailing zeros');
sortCase('sort_tiny', [cloud(17, 4096, { tiny: true })], [0.2, 0.3, 0.93, -6.5], null, 'size/alpha cull exercised');
sortCase('sort_two_push', [cloud(18, 1000), cloud(19, 1500)], [0.7, -0.1, 0.70, -6.0], null, 'append via 2 pushes');
sortCase('sort_n65', [cloud(20, 65)], [0, 1, 0, -4.0], null, 'one more than a wavefront');
{ // protocol: sort before any push -> Uint32Array(1) = [0] (index.js:588-590)
  const idx = runWorker([], V_FRONT, null);
  emit('sort_before_push', 'sort_proto', { sorted: idx }, { note: 'sort before push' });
}

// ---------------------------------------------------------------- PACK cases
function runPack(chunks, texW) {
  let total = 0; for (const c of chunks) total += c.byteLength / 32;
  const texH = Math.floor((total - 1) / texW) + 1;
  const posted = [];
  const gl = { TEXTURE_2D: 1, RGBA: 2, FLOAT: 3, RGBA_INTEGER: 4, UNSIGNED_INT: 5, bindTexture() {}, texSubImage2D() {} };
  const self = Object.create(def);
  Object.assign(self, {
    loadedVertexCount: 0, maxVertexes: texW * texH, bufferTextureWidth: texW, bufferTextureHeight: texH,
    centerAndScaleData: new Float32Array(texW * texH * 4), covAndColorData: new Uint32Array(texW * texH * 4),
    centerAndScaleTexture: {}, covAndColorTexture: {},
    renderer: { getContext: () => gl, properties: { get: () => ({ __webglTexture: {} }) } },
    worker: { postMessage: (m) => { posted.push(new Float32Array(m.matrices)); } },
  });
  for (const c of chunks) self.pushDataBuffer(c, c.byteLength / 32);
  const mats = new Float32Array(total * 16); let o = 0;
  for (const p of posted) { mats.set(p, o); o += p.length; }
  return { cs: self.centerAndScaleData.slice(0, total * 4), cc: self.covAndColorData.slice(0, total * 4), mats, total };
}
function splatRows(seed, n, special) {
  const r = rng(seed); const buf = new ArrayBuffer(n * 32); const f = new Float32Array(buf); const u = new Uint8Array(buf);
  for (let i = 0; i < n; i++) {
    f[i * 8 + 0] = 2.5 * r.n(); f[i * 8 + 1] = 1.0 * r.n(); f[i * 8 + 2] = 2.5 * r.n();
    for (let k = 0; k < 3; k++) f[i * 8 + 3 + k] = Math.exp(Math.min(-1, Math.max(-7, -4.2 + 0.9 * r.n())));
    for (let k = 0; k < 4; k++) u[i * 32 + 24 + k] = Math.floor(256 * r.u());
    let q = [r.n(), r.n(), r.n(), r.n()]; const l = Math.hypot(q[0], q[1], q[2], q[3]);
    for (let k = 0; k < 4; k++) u[i * 32 + 28 + k] = Math.max(0, Math.min(255, Math.round(q[k] / l * 128 + 128)));
  }
  if (special) special(f, u, n);
  return buf;
}
function special(f, u, n) {
  // row 0: identity rotation, needle scale -> tiny covariance ratios: parseInt exponent-form quirk (index.js:386)
  f[3] = 1.0; f[4] = 1e-6; f[5] = 3e-7; u[28] = 255; u[29] = 128; u[30] = 128; u[31] = 128;
  // row 1: zero scale -> max_value 0 -> NaN -> stored 0
  f[8 + 3] = 0; f[8 + 4] = 0; f[8 + 5] = 0;
  // row 2: quaternion bytes at the extremes (un-normalised)
  u[64 + 28] = 0; u[64 + 29] = 255; u[64 + 30] = 0; u[64 + 31] = 255;
  // row 3: all-128 quaternion = zero quaternion -> R = I
  u[96 + 28] = 128; u[96 + 29] = 128; u[96 + 30] = 128; u[96 + 31] = 128;
  // row 4: identity rotation, moderately thin -> exact zeros off-diagonal, small diagonal ratio
  f[32 + 3] = 0.5; f[32 + 4] = 2e-4; f[32 + 5] = 1e-5; u[128 + 28] = 255; u[128 + 29] = 128; u[128 + 30] = 128; u[128 + 31] = 128;
  // row 5: rotated needle -> tiny but non-zero off-diagonals
  f[40 + 3] = 1.0; f[40 + 4] = 1e-5; f[40 + 5] = 1e-5; u[160 + 28] = 254; u[160 + 29] = 129; u[160 + 30] = 128; u[160 + 31] = 128;
  // row 6: huge + denormal scales
  f[48 + 3] = 1e18; f[48 + 4] = 1e-30; f[48 + 5] = 1e-44; u[192 + 28] = 200; u[192 + 29] = 100; u[192 + 30] = 60; u[192 + 31] = 180;
  // row 7: negative scale value (never produced by loaders but representable)
  f[56 + 3] = -0.02;
}
{
  const a = splatRows(31, 300, special);
  const r = runPack([a], 64);
  emit('pack_n300', 'pack', { rows: new Uint8Array(a), center_scale: r.cs, cov_color: r.cc, matrices: r.mats },
    { n: r.total, pushes: [300], note: 'special rows 0..7, see gen_golden.js' });
  const b1 = splatRows(32, 37), b2 = splatRows(33, 100), b3 = splatRows(34, 1);
  const r2 = runPack([b1, b2, b3], 16);
  const cat = new Uint8Array(138 * 32); cat.set(new Uint8Array(b1), 0); cat.set(new Uint8Array(b2), 37 * 32); cat.set(new Uint8Array(b3), 137 * 32);
  emit('pack_append3', 'pack', { rows: cat, center_scale: r2.cs, cov_color: r2.cc, matrices: r2.mats },
    { n: r2.total, pushes: [37, 100, 1], note: 'progressive append, 3 pushes (texture width 16 -> multi-rect upload path)' });
}

// ---------------------------------------------------------------- PLY cases
function plyBytes(props, n, fill, opt) {
  opt = opt || {};
  const SZ = { double: 8, int: 4, uint: 4, float: 4, short: 2, ushort: 2, uchar: 1, char: 1 };
  let hdr = 'ply\nformat binary_little_endian 1.0\n' + (opt.comment ? 'comment ' + opt.comment + '\n' : '') +
    'element vertex ' + n + '\n';
  let row = 0; for (const p of props) { hdr += 'property ' + p[0] + ' ' + p[1] + '\n'; row += SZ[p[0]]; }
  hdr += opt.noEnd ? '' : 'end_header\n';
  const h = Buffer.from(hdr, 'ascii'); const body = Buffer.alloc(row * n); const dv = new DataView(body.buffer, body.byteOffset, body.length);
  for (let i = 0; i < n; i++) { let o = i * row; for (const p of props) { const v = fill(i, p[1]);
    switch (p[0]) { case 'double': dv.setFloat64(o, v, true); break; case 'float': dv.setFloat32(o, v, true); break;
      case 'int': dv.setInt32(o, v, true); break; case 'uint': dv.setUint32(o, v, true); break;
      case 'short': dv.setInt16(o, v, true); break; case 'ushort': dv.setUint16(o, v, true); break;
      case 'uchar': dv.setUint8(o, v); break; case 'char': dv.setInt8(o, v); break; }
    o += SZ[p[0]]; } }
  return Buffer.concat([h, body]);
}
function toAB(b) { return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength); }
function inriaProps() {
  const p = [['float', 'x'], ['float', 'y'], ['float', 'z'], ['float', 'nx'], ['float', 'ny'], ['float', 'nz'],
    ['float', 'f_dc_0'], ['float', 'f_dc_1'], ['float', 'f_dc_2']];
  for (let k = 0; k < 45; k++) p.push(['float', 'f_rest_' + k]);
  p.push(['float', 'opacity']); for (let k = 0; k < 3; k++) p.push(['float', 'scale_' + k]);
  for (let k = 0; k < 4; k++) p.push(['float', 'rot_' + k]);
  return p;
}
function plyCase(name, props, n, fill, note, opt) {
  const b = plyBytes(props, n, fill, opt);
  const out = new Uint8Array(def.processPlyBuffer(toAB(b)));
  emit(name, 'ply', { ply: new Uint8Array(b), rows: out }, { n, note });
}
{
  const r = rng(41); const cache = {};
  const gauss = (i, nm) => { const k = i + ':' + nm; if (!(k in cache)) {
    let v; if (nm === 'opacity') v = 0.5 + 2.5 * r.n(); else if (nm.startsWith('scale_')) v = -4.2 + 0.9 * r.n();
    else if (nm.startsWith('f_dc_')) v = 1.2 * r.n(); else if (nm.startsWith('rot_')) v = r.n(); else v = 2.0 * r.n();
    cache[k] = v; } return cache[k]; };
  plyCase('ply_inria64', inriaProps(), 64, gauss, 'INRIA layout: 62 float props, 248 B/row');
  // tie-heavy importance: only 4 distinct (scale,opacity) classes; f_dc drives .5 rounding (Uint8ClampedArray half-even)
  const tie = (i, nm) => { if (nm === 'opacity') return (i % 2) ? 0.0 : 2.0; if (nm.startsWith('scale_')) return (i % 4 < 2) ? -3.0 : -5.0;
    if (nm.startsWith('f_dc_')) return ((i % 7) - 3) / (0.28209479177387814 * 255) * ((i % 3) + 0.5); if (nm === 'rot_0') return (i % 5) - 2;
    if (nm.startsWith('rot_')) return ((i * 7 + nm.charCodeAt(4)) % 9) - 4 + 0.5; return i * 0.25 - 8; };
  plyCase('ply_ties96', inriaProps(), 96, tie, 'tie-heavy importance -> comparator-sort stability; clamped-array rounding',
    { comment: 'tie heavy' });
  const r2 = rng(42);
  const col = (i, nm) => (nm === 'red' || nm === 'green' || nm === 'blue') ? Math.floor(256 * r2.u()) : 3.0 * r2.n();
  plyCase('ply_color_only', [['float', 'x'], ['float', 'y'], ['float', 'z'], ['uchar', 'red'], ['uchar', 'green'], ['uchar', 'blue']],
    50, col, 'no scale_0: default scale 0.01, rot [255,0,0,0], alpha 255, identity order');
  const r3 = rng(43);
  const mixed = (i, nm) => { if (nm === 'opacity') return 1.5 * r3.n(); if (nm.startsWith('scale_')) return -4 + r3.n();
    if (nm === 'label') return Math.floor(65536 * r3.u()); if (nm === 'flag') return Math.floor(200 * r3.u()) - 100;
    if (nm.startsWith('f_dc_')) return 3 * r3.n(); if (nm.startsWith('rot_')) return Math.floor(2000 * r3.u()) - 1000; return 2 * r3.n(); };
  plyCase('ply_mixed_types', [['double', 'x'], ['double', 'y'], ['float', 'z'], ['ushort', 'label'], ['char', 'flag'],
    ['float', 'f_dc_0'], ['float', 'f_dc_1'], ['float', 'f_dc_2'], ['float', 'opacity'], ['float', 'scale_0'], ['float', 'scale_1'],
    ['float', 'scale_2'], ['short', 'rot_0'], ['short', 'rot_1'], ['int', 'rot_2'], ['float', 'rot_3']], 40, mixed,
  'double/short/int/ushort/char(getInt8 fallback) property types (index.js:613-631)');
  // error behaviour (index.js:606-607, 643)
  const errs = {};
  try { def.processPlyBuffer(toAB(plyBytes([['float', 'x']], 2, () => 0, { noEnd: true }))); } catch (e) { errs.no_end_header = e.message; }
  try { def.processPlyBuffer(toAB(plyBytes([['float', 'x'], ['float', 'y'], ['float', 'z'], ['float', 'scale_0'], ['float', 'scale_1'],
    ['float', 'scale_2'], ['float', 'opacity'], ['float', 'rot_0'], ['float', 'rot_1'], ['float', 'rot_2']], 2, () => 0.5))); } catch (e) { errs.missing_rot_3 = e.message; }
  try { def.processPlyBuffer(toAB(plyBytes([['float', 'x'], ['float', 'y'], ['float', 'z']], 2, () => 0.5))); } catch (e) { errs.missing_red = e.message; }
  manifest.ply_errors = { kind: 'ply_errors', arrays: {}, meta: errs };
}

// ---------------------------------------------------------------- CAMERA / uniform cases
function compose(p, yawDeg, s) {
  const h = yawDeg * Math.PI / 360; const q = new THREE.Quaternion(0, Math.sin(h), 0, Math.cos(h));
  return new THREE.Matrix4().compose(new THREE.Vector3(p[0], p[1], p[2]), q, new THREE.Vector3(s[0], s[1], s[2]));
}
function cameraCase(name, cam, obj, cutout, vpW, vpH, proj, note) {
  const posted = [];
  const self = Object.create(def);
  Object.assign(self, { camera: { matrixWorld: cam, projectionMatrix: proj }, object: { matrixWorld: obj }, sortReady: true,
    worker: { postMessage: (m) => posted.push(m) } });
  if (cutout) self.cutout = { matrixWorld: cutout };
  self.tick(0, 0);
  const view = new Float32Array(posted[0].view); const co = posted[0].cutout;
  const gsMV = self.getModelViewMatrix(); const gsP = self.getProjectionMatrix();
  // uniforms exactly as material.onBeforeRender computes them (index.js:184-195)
  const focal = (vpH / 2.0) * Math.abs(gsP.elements[5]);
  const arrays = { cam_world: new Float64Array(cam.elements), obj_world: new Float64Array(obj.elements), proj: new Float64Array(proj.elements),
    gs_mv: new Float64Array(gsMV.elements), gs_proj: new Float64Array(gsP.elements), view: view,
    viewport: new Float64Array([vpW, vpH]), focal: new Float64Array([focal]) };
  if (cutout) { arrays.cutout_world = new Float64Array(cutout.elements); arrays.cutout = new Float32Array(co); }
  emit(name, 'camera', arrays, { note, has_cutout: !!cutout });
}
{
  const P = (w, h) => new THREE.Matrix4().makePerspectiveFov(80, w / h, 0.005, 10000);
  cameraCase('cam_index_html', compose([0, 1.6, 0], 0, [1, 1, 1]), compose([0, 1.5, -2], 0, [1, 1, 1]), null, 1920, 1080, P(1920, 1080),
    'index.html:13 pose, A-Frame default camera (fov 80, near .005, far 10000) at (0,1.6,0)');
  cameraCase('cam_index_yaw37', compose([0, 1.6, 0], 0, [1, 1, 1]), compose([0, 1.5, -2], 37, [1, 1, 1]), null, 1920, 1080, P(1920, 1080),
    'index.html pose, entity yaw 37 deg (orbit frame)');
  cameraCase('cam_cutout_demo', compose([5.132, 1.6, 7.237], 0, [1, 1, 1]), compose([0, 0.8, -2], 0, [2, 2, 2]),
    compose([0.8145, 1.73322, -2.35981], 0, [4.17, 2.95, 3.89]), 1280, 720, P(1280, 720), 'cutout-demo.html:22-24');
  cameraCase('cam_yawed_camera', compose([1.0, 1.2, 3.0], -25, [1, 1, 1]), compose([0.3, 1.5, -2], 110, [1.5, 1.5, 1.5]),
    compose([0.2, 1.0, -2.0], 30, [2, 3, 2]), 1032, 1104, new THREE.Matrix4().makePerspective(-0.006, 0.0045, 0.005, -0.0052, 0.005, 10000),
    'rotated camera + rotated/scaled entity + rotated cutout, asymmetric (XR-like) frustum');
}

// ---------------------------------------------------------------- Math.exp of the engine the reference runs on
// processPlyBuffer's importance, scale and opacity all go through Math.exp (index.js:659-662, 700, 722): V8's
// ieee754::exp (fdlibm e_exp.c), not a correctly-rounded exp.  These vectors pin the restatements (oracle gso_js_exp,
// product gsm::js_exp) to what THIS engine returns, bit for bit.
{
  const r = rng(77); const xs = [];
  const f32 = (v) => Math.fround(v);
  for (const v of [0, -0, 1, -1, 0.5, -0.5, Infinity, -Infinity, NaN, 709.782712893384, 709.7827128933841, 710, -745.1332191019411,
    -745.1332191019412, -746, -708.3964185322641, -720, 1e-10, -1e-10, 3.7252902984619140625e-9, 0.34657359027997264, 0.3465735902799727,
    1.0397207708399179, 1.039720770839918, -0.34657359027997264, -1.0397207708399179, 88.72283905206835, -87.33654475055310898657, 700.5, -700.5]) xs.push(v);
  for (let i = 0; i < 3000; i++) xs.push(f32(-4.2 + 0.9 * r.n()));              // ln-scale values as they sit in a PLY (f32)
  for (let i = 0; i < 3000; i++) xs.push(-f32(0.5 + 2.5 * r.n()));              // -opacity
  for (let i = 0; i < 1500; i++) xs.push(f32(60 * r.u() - 40));                  // wide f32 range
  for (let i = 0; i < 600; i++) xs.push((r.u() - 0.5) * 1.5);                    // around the reduction thresholds, full f64 mantissas
  for (let i = 0; i < 70; i++) xs.push(-700 - 50 * r.u());                       // gradual underflow
  const x = new Float64Array(xs), y = new Float64Array(xs.length);
  for (let i = 0; i < x.length; i++) y[i] = Math.exp(x[i]);
  emit('math_exp', 'math', { x, exp: y }, { note: 'Math.exp under ' + process.version + ' (V8 ' + process.versions.v8 + ')' });
}
// a larger PLY so that the importance order is a real sort (4096 rows, compact 14-float layout)
{
  const r = rng(91);
  const g = (i, nm) => { if (nm === 'opacity') return 0.5 + 2.5 * r.n(); if (nm.startsWith('scale_')) return -4.2 + 0.9 * r.n();
    if (nm.startsWith('f_dc_')) return 1.2 * r.n(); if (nm.startsWith('rot_')) return r.n(); return 2.0 * r.n(); };
  const props = ['x', 'y', 'z', 'f_dc_0', 'f_dc_1', 'f_dc_2', 'opacity', 'scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3'].map((n) => ['float', n]);
  plyCase('ply_n4096', props, 4096, g, 'compact 14-float layout, 4096 rows: importance order of a real-sized sort');
}

fs.writeFileSync(path.join(OUT, 'manifest.json'), JSON.stringify(manifest, null, 1));
let bytes = 0; for (const f of fs.readdirSync(OUT)) bytes += fs.statSync(path.join(OUT, f)).size;
console.log('wrote', Object.keys(manifest).length, 'cases,', bytes, 'bytes ->', OUT);
