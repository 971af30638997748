// gaussian_splatting.js -- host side of the drop-in, in the reference's own language.
//
// A headless mirror of the A-Frame `gaussian_splatting` component surface (reference index.js:1-23 schema/init,
// 222 loadData, 328 pushDataBuffer, 438 tick, 456/467 camera matrices, 488 createWorker, 600 processPlyBuffer):
// same property names and defaults, same method names, same worker message protocol -- but the Worker sort and
// the WebGL draw are replaced by the MI355X HIP path behind the N-API addon (gs_splat_napi.node -> libgs_splat_hip.so),
// and `render()` returns the RGBA framebuffer that three.js would have drawn.
//
// camera / object arguments are duck-typed like three.js objects: {matrixWorld: {elements[16]},
// projectionMatrix: {elements[16]}} (column-major).  There is no JS fallback for the hot path.
'use strict';
const fs = require('fs');
const native = require('./gs_splat_napi.node');

const schema = {                                   // index.js:2-7
  src: { type: 'string', default: 'train.splat' },
  cutoutEntity: { type: 'selector' },
  pixelRatio: { type: 'number', default: 1 },
  xrPixelRatio: { type: 'number', default: 0.5 },
};
const ROW_LENGTH = 3 * 4 + 3 * 4 + 4 + 4;          // index.js:227

function elementsOf(m) { return m && m.elements ? m.elements : m; }

class GaussianSplatting {
  constructor(data, options) {
    this.data = Object.assign({ src: schema.src.default, cutoutEntity: null, pixelRatio: schema.pixelRatio.default,
      xrPixelRatio: schema.xrPixelRatio.default }, data || {});
    this.device = (options && options.device) || 0;
    this.handle = native.create(this.device);       // replaces `new Worker(...)` + GL resource creation
    this.loadedVertexCount = 0;
    this.rowLength = ROW_LENGTH;
    this.sortReady = true;
    this.instanceCount = 0;
    this.sortedIndexes = null;
    this.cutout = null;
  }

  // init (index.js:8-23): resolution properties are only applied when > 0; the cutout entity is resolved to its object3D.
  init(sceneEl) {
    const r = sceneEl && sceneEl.renderer;
    if (r && this.data.pixelRatio > 0 && r.setPixelRatio) r.setPixelRatio(this.data.pixelRatio);
    if (r && this.data.xrPixelRatio > 0 && r.xr && r.xr.setFramebufferScaleFactor) r.xr.setFramebufferScaleFactor(this.data.xrPixelRatio);
    if (this.data.cutoutEntity) this.cutout = this.data.cutoutEntity.object3D || this.data.cutoutEntity;
    return this;
  }

  // drawing-buffer size the reference would get from renderer.setPixelRatio / xr.setFramebufferScaleFactor
  framebufferSize(cssWidth, cssHeight, xr) {
    const [w, h] = native.scaledSize(cssWidth, cssHeight, xr ? this.data.xrPixelRatio : this.data.pixelRatio);
    return { width: w, height: h };
  }

  // loadData (index.js:222-327).  No network on the target: `src` is a path or file:// URL.  `.splat` files are
  // ingested progressively chunk by chunk exactly like the streaming branch (index.js:279-298); `.ply` is read whole,
  // converted by processPlyBuffer, then pushed (index.js:305-325).
  async loadData(camera, object, renderer, src, chunkBytes) {
    this.camera = camera; this.object = object; this.renderer = renderer;
    this.loadedVertexCount = 0;
    native.clear(this.handle);                       // worker.postMessage({method: "clear"})
    const file = String(src || this.data.src).replace(/^file:\/\//, '');
    const isPly = file.endsWith('.ply');
    const total = fs.statSync(file).size;
    const fd = fs.openSync(file, 'r');
    try {
      if (isPly) {
        const all = Buffer.alloc(total); fs.readSync(fd, all, 0, total, 0);
        const rows = this.processPlyBuffer(all.buffer.slice(all.byteOffset, all.byteOffset + total));
        this.pushDataBuffer(rows, Math.floor(rows.byteLength / this.rowLength));
      } else {
        const step = chunkBytes || (1 << 22);
        let pending = Buffer.alloc(0), pos = 0;
        while (pos < total) {
          const buf = Buffer.alloc(Math.min(step, total - pos));
          fs.readSync(fd, buf, 0, buf.length, pos); pos += buf.length;
          pending = pending.length ? Buffer.concat([pending, buf]) : buf;
          if (pending.length > this.rowLength) {     // bytesRemains > rowLength (index.js:280)
            const vertexCount = Math.floor(pending.length / this.rowLength), used = vertexCount * this.rowLength;
            const ab = pending.buffer.slice(pending.byteOffset, pending.byteOffset + used);
            this.pushDataBuffer(ab, vertexCount);
            pending = pending.slice(used);
          }
        }
        if (pending.length >= this.rowLength) {
          const n = Math.floor(pending.length / this.rowLength);
          this.pushDataBuffer(pending.buffer.slice(pending.byteOffset, pending.byteOffset + n * this.rowLength), n);
        }
      }
    } finally { fs.closeSync(fd); }
    this.sortReady = true;
    return this.loadedVertexCount;
  }

  // pushDataBuffer (index.js:328-437): pack + upload + worker push, all on the GPU side now.
  pushDataBuffer(buffer, vertexCount) {
    if (vertexCount <= 0) return;
    native.pushSplat(this.handle, buffer, vertexCount);
    this.loadedVertexCount += vertexCount;
  }

  // tick (index.js:438-455) + the worker reply handler (index.js:201-207), single flight.  The reference always sorts
  // from this.camera (in XR: the head camera, one order for both eyes); passing an eye camera sorts for that eye alone
  // (SURVEY.md 8f-3): comp.tick(eye); comp.render(eye, viewport).
  tick(camera) {
    if (!this.sortReady) return;
    this.sortReady = false;
    try {
      const u = this._tickUniforms(camera);
      const indexes = native.sort(this.handle, u.view, u.cutout);
      this.sortedIndexes = indexes;
      this.instanceCount = indexes.length;
    } finally { this.sortReady = true; }
  }

  // The reference's rhythm: tick posts the sort and returns; the reply handler installs the new order and re-arms sortReady
  // (index.js:201-207, 438-455) -- and every frame drawn meanwhile uses the last COMPLETED order.  Returns the promise of the order,
  // or null while a sort is in flight.  render() / frame-less draws stay callable in between (they draw the old order); the new
  // order is installed when the promise resolves.  No second thread: gs_sort_begin enqueues the sort on a pipeline lane of its own,
  // the promise polls it (gs_sort_poll) from the event loop.
  tickAsync(camera) {
    if (!this.sortReady) return null;
    this.sortReady = false;
    const u = this._tickUniforms(camera);
    try { native.sortBegin(this.handle, u.view, u.cutout); } catch (e) { this.sortReady = true; throw e; }
    return new Promise((resolve, reject) => {
      const poll = () => {
        if (this.sortReady) { resolve(this.sortedIndexes); return; }            // (collected meanwhile by tickFinish)
        let indexes;
        try { indexes = native.sortPoll(this.handle, false, true); } catch (e) { this.sortReady = true; reject(e); return; }
        if (indexes === null) { setImmediate(poll); return; }
        this.sortedIndexes = indexes; this.instanceCount = indexes.length; this.sortReady = true;
        resolve(indexes);
      };
      setImmediate(poll);
    });
  }

  // the same reply collected at once (blocks until the posted sort has run): what a caller without an event loop turn to spare does
  tickFinish() {
    if (this.sortReady) return this.sortedIndexes;
    const indexes = native.sortPoll(this.handle, true, true);
    this.sortedIndexes = indexes; this.instanceCount = indexes.length; this.sortReady = true;
    return indexes;
  }

  _tickUniforms(camera) {
    return native.tickUniforms(elementsOf((camera || this.camera).matrixWorld), elementsOf(this.object.matrixWorld),
      this.cutout ? elementsOf(this.cutout.matrixWorld) : undefined);
  }

  getProjectionMatrix(camera) {                      // index.js:456-466
    if (!camera) camera = this.camera;
    return { elements: native.projectionMatrix(elementsOf(camera.projectionMatrix)) };
  }

  getModelViewMatrix(camera) {                       // index.js:467-487
    if (!camera) camera = this.camera;
    return { elements: native.modelViewMatrix(elementsOf(camera.matrixWorld), elementsOf(this.object.matrixWorld)) };
  }

  // The draw: onBeforeRender uniforms (index.js:184-195) + vertex/fragment/blend (index.js:77-181) -> RGBA8 pixels.
  // viewport = {width, height[, x0, x1]} in device pixels; returns Uint8Array, row 0 = top.
  // The pixels land in page-locked memory owned by the component and reused frame after frame (one buffer per strip size):
  // the returned Uint8Array is overwritten by the next render of the same size -- copy it to keep it.
  render(camera, viewport, options) {
    const p = this._renderParams(camera, viewport, options);
    return native.renderInto(this.handle, p, this._frame(p));
  }

  // One whole frame, synchronously, for a caller that wants pixels and not the index list: this frame's sort (the order stays on
  // the GPU -- tick() hands sortedIndexes back to JavaScript like the reference's worker does, ~3 MB per tick at 1 M splats) and
  // the draw.  Returns the frame like render().
  frame(camera, viewport, options) {
    const u = this._tickUniforms(camera);
    native.sort(this.handle, u.view, u.cutout, false);
    const p = this._renderParams(camera, viewport, options);
    return native.renderInto(this.handle, p, this._frame(p));
  }

  // The same draw off the JS thread (napi_async_work): resolves to the frame.  While it is in flight the component's
  // other calls throw GS_BUSY (a context is single-caller, like the reference's single-flight worker).
  renderAsync(camera, viewport, options) {
    const p = this._renderParams(camera, viewport, options);
    return native.renderAsync(this.handle, p, this._frame(p));
  }

  _renderParams(camera, viewport, options) {
    const proj = this.getProjectionMatrix(camera).elements;
    const p = Object.assign({
      modelView: this.getModelViewMatrix(camera).elements, projection: proj,
      width: viewport.width, height: viewport.height,
      focal: (viewport.height / 2.0) * Math.abs(proj[5]),
    }, options || {});
    if (viewport.x0 !== undefined) p.x0 = viewport.x0;
    if (viewport.x1 !== undefined) p.x1 = viewport.x1;
    return p;
  }

  _frame(p, slot) {
    const w = (p.x1 !== undefined ? p.x1 : p.width) - (p.x0 || 0), key = w + 'x' + p.height + (slot ? '#' + slot : '');
    if (!this._frames) this._frames = new Map();
    if (!this._frames.has(key)) this._frames.set(key, native.allocFrame(w, p.height));
    return this._frames.get(key);
  }

  // Throughput mode -- what WebGL does behind the reference's back: the draw call returns at once and frames queue up on the
  // GPU (index.js:184-207).  frameQueued = this frame's sort (the order stays on the GPU) + draw, enqueued on one of the
  // library's pipeline lanes (two frames per launch, GS_OPT_FRAME_BATCH); the pixels follow their kernels into one of
  // `QUEUE_DEPTH` page-locked frames, which is what is
  // returned -- valid after sync().  A frame that comes back incomplete (a buffer had to grow, a skipped binning round was
  // needed after all) is drawn again by sync() itself into the same frame (GS_OPT_AUTO_RETRY); sync() throws code GS-9
  // (GS_E_RETRY) only if more than QUEUE_DEPTH frames were queued between two sync() calls (frames sharing a buffer).
  frameQueued(camera, viewport, options) {
    if (!this._paired) { native.setOption(this.handle, 10, 2); this._paired = true; }   // GS_OPT_FRAME_BATCH: consecutive queued frames share their launches
    const u = this._tickUniforms(camera);
    native.sort(this.handle, u.view, u.cutout, false);
    const p = this._renderParams(camera, viewport, options);
    p.flags = (p.flags || 0) | 8;                      // GS_RENDER_ASYNC
    this._slot = ((this._slot || 0) % GaussianSplatting.QUEUE_DEPTH) + 1;
    return native.renderInto(this.handle, p, this._frame(p, this._slot));
  }

  sync() { native.sync(this.handle); }

  // The opaque scene three.js draws before the transparent splat mesh: window-space depth (depthTest: true,
  // depthWrite: false, index.js:179-180) and colour.  Float32Array / Uint8Array of width*height(*4), row 0 = top.
  setScene(depth, rgba, width, height) { native.setScene(this.handle, depth || null, rgba || null, width || 0, height || 0); }

  // createWorker (index.js:488-599): same message protocol, GPU-backed.  `self` needs postMessage; onmessage is installed.
  createWorker(self) {
    const h = native.create(this.device);
    let havePush = false;
    self.onmessage = (e) => {
      const d = e.data;
      if (d.method === 'clear') { native.clear(h); havePush = false; }
      if (d.method === 'push') { native.pushMatrices(h, d.matrices); havePush = true; }
      if (d.method === 'sort') {
        const sortedIndexes = havePush ? native.sort(h, d.view, d.cutout) : new Uint32Array(1);   // index.js:588-590
        self.postMessage({ sortedIndexes }, [sortedIndexes.buffer]);
      }
    };
    return self;
  }

  // processPlyBuffer (index.js:600-745): header on the host, importance order + row conversion on the GPU
  processPlyBuffer(inputBuffer) { return native.plyToSplatGpu(this.handle, inputBuffer); }

  stats() { return native.stats(this.handle); }

  // the reference leaks its worker and textures; this frees the context (HBM, streams, threads) at once
  remove() { native.destroy(this.handle); this._frames = null; }
}

GaussianSplatting.QUEUE_DEPTH = 7;                // frames to queue between two sync() calls in throughput mode: the first goes out alone
                                                   // at once, then GS_OPT_PIPELINE_DEPTH's default (3) pairs

// The registered component (index.js:1-23): AFRAME.registerComponent("gaussian_splatting", ...) with the reference's life cycle.
//   init    resolution properties (index.js:10-15); when the scene has loaded (index.js:17), loadData with the scene camera's
//           three.js camera, the entity's object3D, the renderer and `src` (index.js:18), and cutoutEntity -> its object3D
//           (index.js:19-21).  `this.ready` is the promise of that load.
//   tick    the reference's tick posts the worker sort (index.js:438-455) and three.js then draws the mesh; here tick sorts
//           and, if the host installed a frame sink, draws: `component.frameSink = (rgba, viewport) => ...` receives the
//           RGBA framebuffer three.js would have drawn (page-locked, reused: copy it to keep it).  The drawing-buffer size is
//           the renderer's (getDrawingBufferSize / domElement), i.e. what pixelRatio / xrPixelRatio made of the canvas.
//   remove  frees the context.
function sceneCamera(sceneEl) {                      // this.el.sceneEl.camera.el.components.camera.camera (index.js:18)
  const c = sceneEl && sceneEl.camera;
  return c && c.el && c.el.components && c.el.components.camera ? c.el.components.camera.camera : c;
}
function drawingBuffer(renderer) {
  if (!renderer) return null;
  if (renderer.getDrawingBufferSize) { const v = renderer.getDrawingBufferSize({ x: 0, y: 0, set(x, y) { this.x = x; this.y = y; return this; } });
    const w = v.width !== undefined ? v.width : v.x, h = v.height !== undefined ? v.height : v.y; if (w > 0 && h > 0) return { width: w, height: h }; }
  const el = renderer.domElement;
  return el && el.width > 0 && el.height > 0 ? { width: el.width, height: el.height } : null;
}
function register(AFRAME) {
  AFRAME.registerComponent('gaussian_splatting', {
    schema,
    init() {
      const sceneEl = this.el && this.el.sceneEl;
      this.impl = new GaussianSplatting(Object.assign({}, this.data, { cutoutEntity: null })).init(sceneEl);
      this.ready = null;
      const start = () => {
        if (this.data.cutoutEntity) this.impl.cutout = this.data.cutoutEntity.object3D;          // index.js:19-21
        this.ready = this.impl.loadData(sceneCamera(sceneEl), this.el.object3D, sceneEl.renderer, this.data.src);   // index.js:18
        return this.ready;
      };
      if (!sceneEl) return;
      if (sceneEl.hasLoaded) start(); else sceneEl.addEventListener('loaded', start);               // index.js:17
    },
    tick() {
      const g = this.impl;
      if (!g || !g.camera || !g.loadedVertexCount) return;
      g.tick();                                                                                     // index.js:438-455
      if (this.frameSink) {
        const vp = this.viewport || drawingBuffer(g.renderer);
        if (vp) this.frameSink(g.render(g.camera, vp), vp);
      }
    },
    remove() { if (this.impl) this.impl.remove(); this.impl = null; },
  });
}

module.exports = { schema, GaussianSplatting, register, native };
