#!/usr/bin/env node
// TEST INFRASTRUCTURE (oracle side). Not part of the product path.
//
// Pixel goldens from the reference's OWN shaders.  Runs the reference component (/root/reference/index.js, read by absolute
// path, never copied) under node with recording stand-ins for the three.js objects it builds in initGL (index.js:26-221):
// the ShaderMaterial's vertex / fragment shader text, blending and depth flags, the quad's vertex positions, the two data
// textures that pushDataBuffer fills, the instanced index attribute that the worker's reply fills, and the uniforms that
// material.onBeforeRender computes -- everything the WebGL draw consumes.  That draw is then executed by oracle/_ref/gl_ref
// (oracle/gl_ref.c: Mesa llvmpipe through the DRI swrast interface, no X server needed) and its framebuffers are stored as
// fixtures: tests/golden/gl_<case>.bin.gz + manifest_gl.json.  A fixture holds DATA only (scene rows, camera, the reference's
// sorted order and uniforms, pixels); the shader text goes to a scratch directory under oracle/_ref/ (git-ignored) and is
// never stored in the repository.
//
// Runs only in the build container (needs /root/reference and Mesa's swrast_dri.so).  On the GPU box the committed fixtures
// are used and this script is never executed.
//
//   python oracle/make_gl_scenes.py      # scene rows + cameras -> oracle/_ref/gl_scenes/ (same generator as the benchmark's)
//   node oracle/gen_golden_gl.js         # regenerate tests/golden/gl_*.bin.gz
'use strict';
const fs = require('fs');
const path = require('path');
const vm = require('vm');
const { execFileSync } = require('child_process');
const crypto = require('crypto');
const zlib = require('zlib');
const T = require('./three_standin.js');

const REF = '/root/reference/index.js';
const OUT = process.env.GS_GL_OUT || path.join(__dirname, '..', 'tests', 'golden');   // (GS_GL_OUT / GS_GL_ONLY: side runs, e.g. timing a whole frame)
const SCRATCH = path.join(__dirname, '_ref');
const GLREF = path.join(SCRATCH, 'gl_ref');
const SCENES = path.join(SCRATCH, 'gl_scenes');
if (!fs.existsSync(REF)) { console.error('reference not present; nothing to do'); process.exit(0); }
if (!fs.existsSync(GLREF)) { console.error('oracle/_ref/gl_ref not built (python __graft_entry__.py)'); process.exit(1); }

// ---------------------------------------------------------------- recording stand-ins for what initGL constructs
class DataTexture { constructor(data, width, height, format, type) { Object.assign(this, { data, width, height, format, type, needsUpdate: false }); } }
class BufferAttribute {
  constructor(array, itemSize) { this.array = array; this.itemSize = itemSize; this.needsUpdate = false; }
  setXYZ(i, x, y, z) { this.array[i * 3] = x; this.array[i * 3 + 1] = y; this.array[i * 3 + 2] = z; return this; }
  set(values) { this.array.set(values); return this; }
  setUsage() { return this; }
}
class InstancedBufferAttribute extends BufferAttribute {}
class BufferGeometry {
  constructor() { this.attributes = {}; }
  setAttribute(name, attr) { this.attributes[name] = attr; return this; }
  copy(g) { Object.assign(this.attributes, g.attributes); return this; }
}
class InstancedBufferGeometry extends BufferGeometry { constructor() { super(); this.instanceCount = Infinity; } }
class ShaderMaterial { constructor(p) { Object.assign(this, p); } }
class Mesh { constructor(geometry, material) { this.geometry = geometry; this.material = material; this.frustumCulled = true; } }
const THREE = Object.assign({}, T, {
  DataTexture, BufferAttribute, InstancedBufferAttribute, BufferGeometry, InstancedBufferGeometry, ShaderMaterial, Mesh,
  RGBA: 'RGBA', FloatType: 'Float', RGBAIntegerFormat: 'RGBAInteger', UnsignedIntType: 'UnsignedInt', DynamicDrawUsage: 'Dynamic',
  CustomBlending: 'CustomBlending', OneFactor: 'One',
});

let def;
const quiet = { log() {}, error() {}, time() {}, timeEnd() {} };
const ctx = {
  AFRAME: { registerComponent: (n, d) => { def = d; } }, THREE, console: quiet, TextDecoder, Math, parseInt, Float32Array, Uint8Array,
  Uint32Array, Int16Array, Int32Array, Uint8ClampedArray, DataView, ArrayBuffer, Proxy, Error, Promise, setTimeout, Infinity,
};
vm.createContext(ctx);
vm.runInContext(fs.readFileSync(REF, 'utf8'), ctx, { filename: REF });

// ---------------------------------------------------------------- fixture writer (same container format as gen_golden.js)
const DT = new Map([[Float32Array, 'f4'], [Float64Array, 'f8'], [Uint32Array, 'u4'], [Uint8Array, 'u1']]);
const manifest = (process.env.GS_GL_MERGE && fs.existsSync(path.join(OUT, 'manifest_gl.json'))) ? JSON.parse(fs.readFileSync(path.join(OUT, 'manifest_gl.json'), 'utf8')) : {};   // GS_GL_MERGE: add / replace cases, keep the others
function emit(name, arrays, meta) {
  const chunks = []; let off = 0; const desc = {};
  for (const k of Object.keys(arrays)) {
    const a = arrays[k]; const dt = DT.get(a.constructor);
    if (!dt) throw new Error('dtype? ' + k);
    const b = Buffer.from(a.buffer, a.byteOffset, a.byteLength);
    desc[k] = { dtype: dt, count: a.length, offset: off };
    chunks.push(b); off += b.length;
    const pad = (8 - (off % 8)) % 8; if (pad) { chunks.push(Buffer.alloc(pad)); off += pad; }
  }
  fs.writeFileSync(path.join(OUT, name + '.bin.gz'), zlib.gzipSync(Buffer.concat(chunks), { level: 9 }));   // (images: 3-5x smaller)
  manifest[name] = { kind: 'gl', arrays: desc, meta: meta || {} };
}

// ---------------------------------------------------------------- one case: the reference's own load -> sort -> draw-state path
async function glCase(name, sc) {
  const W = sc.width, H = sc.height;
  const rows = fs.readFileSync(path.join(SCENES, sc.rows));
  const n = rows.length / 32;
  // gl.MAX_TEXTURE_SIZE as this "renderer" reports it: the texture width, and its square the splat capacity (index.js:31-40)
  const TEXW = n > (1 << 24) ? 8192 : (n > (1 << 20) ? 4096 : 1024);
  const M = (e) => { const m = new THREE.Matrix4(); m.elements = Array.from(e); return m; };
  const gl = { MAX_TEXTURE_SIZE: 'MAX', TEXTURE_2D: 1, RGBA: 2, FLOAT: 3, RGBA_INTEGER: 4, UNSIGNED_INT: 5,
    getParameter: () => TEXW, bindTexture() {}, texSubImage2D() {} };
  let mesh = null;
  const self = Object.create(def);
  const workerSelf = { postMessage: (m) => { self.worker.onmessage({ data: m }); } };
  def.createWorker(workerSelf);                                 // the reference's worker code, wired back to back
  Object.assign(self, {
    data: {}, loadedVertexCount: 0, rowLength: 32, sortReady: false,
    camera: { matrixWorld: M(sc.cam_world), projectionMatrix: M(sc.proj) },
    object: { matrixWorld: M(sc.obj_world), add: (m) => { mesh = m; }, frustumCulled: true },
    renderer: { getContext: () => gl, properties: { get: () => ({ __webglTexture: {} }) },
      getCurrentViewport: (v) => { v.x = 0; v.y = 0; v.z = W; v.w = H; return v; } },
    worker: { postMessage: (m) => { workerSelf.onmessage({ data: m }); }, onmessage: null },
  });
  if (sc.cutout_world) self.cutout = { matrixWorld: M(sc.cutout_world) };
  await self.initGL(n);                                          // index.js:26-221: textures, geometry, material, reply handler
  // index.js:328-437: fills both textures, pushes the worker rows -- in one call, or chunk by chunk like a progressive load
  // (index.js:279-298); the worker appends each push to what it holds (index.js:576-586)
  const chunk = sc.push_chunk || n;
  for (let o = 0; o < n; o += chunk) {
    const m = Math.min(chunk, n - o);
    self.pushDataBuffer(rows.buffer.slice(rows.byteOffset + o * 32, rows.byteOffset + (o + m) * 32), m);
  }
  self.sortReady = true;
  self.tick(0, 0);                                               // index.js:438-455 -> worker sort -> reply handler (index.js:201-207)
  const mat = mesh.material, geo = mesh.geometry;
  // (XR: three.js draws each eye with the eye's camera, while tick sorted from this.camera, the head: index.js:441 vs 185-187)
  const drawCamera = sc.eye_cam_world ? { matrixWorld: M(sc.eye_cam_world), projectionMatrix: M(sc.eye_proj) } : self.camera;
  mat.onBeforeRender(self.renderer, null, drawCamera, geo, mesh, null);   // index.js:184-195
  const count = geo.instanceCount;
  const u = mat.uniforms;
  if (mat.blending !== 'CustomBlending' || mat.blendSrcAlpha !== 'One' || mat.blendSrc !== undefined || mat.blendDst !== undefined ||
      mat.blendEquation !== undefined || mat.transparent !== true) throw new Error('material state differs from what gl_ref sets up');

  // ---- the draw, on Mesa
  const job = path.join(SCRATCH, 'gl_job_' + name);
  fs.mkdirSync(job, { recursive: true });
  fs.writeFileSync(path.join(job, 'vs.glsl'), mat.vertexShader);
  fs.writeFileSync(path.join(job, 'fs.glsl'), mat.fragmentShader);
  const f32 = (a) => Buffer.from(new Float32Array(a).buffer);
  fs.writeFileSync(path.join(job, 'positions.bin'), f32(geo.attributes.position.array));
  const idx = new Uint32Array(geo.attributes.splatIndex.array.buffer, 0, count);
  fs.writeFileSync(path.join(job, 'index.bin'), Buffer.from(idx.buffer, idx.byteOffset, Math.max(4, count * 4)));
  const cs = self.centerAndScaleTexture, cc = self.covAndColorTexture;
  fs.writeFileSync(path.join(job, 'cs.bin'), Buffer.from(cs.data.buffer, cs.data.byteOffset, cs.data.byteLength));
  fs.writeFileSync(path.join(job, 'cc.bin'), Buffer.from(cc.data.buffer, cc.data.byteOffset, cc.data.byteLength));
  const fl = (a) => Array.from(new Float32Array(a)).map((v) => v.toPrecision(9)).join(' ');
  fs.writeFileSync(path.join(job, 'job.txt'), [
    'width ' + W, 'height ' + H, 'tex_width ' + cs.width, 'tex_height ' + cs.height, 'instances ' + count,
    'depth_test ' + (mat.depthTest ? 1 : 0), 'depth_write ' + (mat.depthWrite ? 1 : 0),
    'viewport ' + fl(u.viewport.value), 'focal ' + fl([u.focal.value]), 'clear 0 0 0 1', sc.strip ? 'strip ' + sc.strip[0] + ' ' + sc.strip[1] : '',
    'projection ' + fl(u.gsProjectionMatrix.value.elements), 'model_view ' + fl(u.gsModelViewMatrix.value.elements), ''].join('\n'));
  for (const k of ['scene_depth', 'scene_rgba']) if (sc[k]) fs.copyFileSync(path.join(SCENES, sc[k]), path.join(job, k + '.bin'));
  execFileSync(GLREF, [job], { stdio: 'inherit' });
  const info = {};
  for (const line of fs.readFileSync(path.join(job, 'out.txt'), 'utf8').split('\n')) { const k = line.indexOf(' '); if (k > 0) info[line.slice(0, k)] = line.slice(k + 1); }
  const rgba8 = new Uint8Array(fs.readFileSync(path.join(job, 'out_rgba8.bin')));
  const fb = fs.readFileSync(path.join(job, 'out_float.bin'));
  const rgbaf = new Float32Array(fb.buffer, fb.byteOffset, fb.length / 4);
  // the float framebuffer rounded ONCE to RGBA8 (clamp, * 255, + 0.5, truncate): what the reference's shading, raster and
  // blend give without the per-fragment unorm8 rounding of an 8-bit colour buffer
  const rgbaOnce = new Uint8Array(rgbaf.length);
  for (let i = 0; i < rgbaf.length; i++) rgbaOnce[i] = Math.floor(Math.fround(Math.fround(Math.min(Math.max(rgbaf[i], 0), 1) * 255) + 0.5));
  const sceneArrays = {};
  if (sc.scene_depth) { const b = fs.readFileSync(path.join(SCENES, sc.scene_depth)); sceneArrays.scene_depth = Float32Array.from(new Float32Array(b.buffer, b.byteOffset, b.length / 4)); }
  if (sc.scene_rgba) sceneArrays.scene_rgba = new Uint8Array(fs.readFileSync(path.join(SCENES, sc.scene_rgba)));
  // big scenes (the BASELINE sizes) are stored by recipe: the generator call that makes the rows + their SHA-1, the SHA-1 of the
  // reference's sorted order, and a column strip of the frame (drawn with a scissor)
  const sha1 = (b) => crypto.createHash('sha1').update(b).digest('hex');
  const small = sc.store_rows !== false;
  emit('gl_' + name, Object.assign({
    rows: small ? new Uint8Array(rows) : new Uint8Array(0), cam_world: new Float64Array(sc.cam_world), obj_world: new Float64Array(sc.obj_world),
    proj: new Float64Array(sc.proj), cutout_world: new Float64Array(sc.cutout_world || []), sorted: small ? Uint32Array.from(idx) : new Uint32Array(0),
    eye_cam_world: new Float64Array(sc.eye_cam_world || []), eye_proj: new Float64Array(sc.eye_proj || []),
    gs_mv: new Float64Array(u.gsModelViewMatrix.value.elements), gs_proj: new Float64Array(u.gsProjectionMatrix.value.elements),
    viewport: new Float64Array(u.viewport.value), focal: new Float64Array([u.focal.value]),
    rgba8_fb: sc.store_rgba8 === false ? new Uint8Array(0) : rgba8, rgba_float_fb_rounded: rgbaOnce,
  }, sceneArrays), { n, width: W, height: H, strip: sc.strip || [0, W], instances: count, fragments: Number(info.fragments_float),
    fragments_rgba8_fb: Number(info.fragments_rgba8), renderer: info.renderer, gl_version: info.version, has_cutout: !!sc.cutout_world,
    has_scene: !!(sc.scene_depth || sc.scene_rgba), rows_sha1: sha1(rows), sorted_sha1: sha1(Buffer.from(idx.buffer, idx.byteOffset, count * 4)),
    rows_recipe: sc.rows_recipe || null, note: sc.note });
  console.log('gl_' + name + ':', n, 'splats,', count, 'instances,', info.fragments_float, 'fragments,', info.renderer, '| draw', info.draw_seconds_rgba8, 's');
  for (const f of fs.readdirSync(job)) fs.unlinkSync(path.join(job, f));   // the scratch copy of the shader text does not outlive the run
  fs.rmdirSync(job);
}

(async () => {
  const scenes = JSON.parse(fs.readFileSync(path.join(SCENES, 'scenes.json'), 'utf8'));
  for (const name of Object.keys(scenes)) if (!process.env.GS_GL_ONLY || process.env.GS_GL_ONLY === name) await glCase(name, scenes[name]);
  fs.writeFileSync(path.join(OUT, 'manifest_gl.json'), JSON.stringify(manifest, null, 1));
  // the scratch scenes (up to 200 MB of rows) are not needed any more -- and oracle/_ref/ travels to the GPU box
  for (const f of fs.readdirSync(SCENES)) fs.unlinkSync(path.join(SCENES, f));
  fs.rmdirSync(SCENES);
  let bytes = 0; for (const f of fs.readdirSync(OUT)) if (f.startsWith('gl_')) bytes += fs.statSync(path.join(OUT, f)).size;
  console.log('wrote', Object.keys(manifest).length, 'GL cases,', bytes, 'bytes ->', OUT);
})().catch((e) => { console.error('FAILED:', e); process.exit(1); });
