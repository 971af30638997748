// node bench_visible.js <scene.splat> <width> <height> [frames = 240] [queueDepth = 48]
// The frame rate a JavaScript caller of the component sees at a given size (bench.py's `js_visible`, north_star: "returns an RGBA
// framebuffer to JavaScript"): a moving entity (a new pose every frame, index.js:438-455), every frame = sort + draw.
//   sync    comp.frame(camera, viewport)                          one frame at a time into the component's page-locked frame, the order
//           stays on the GPU;  tick_render = comp.tick(); comp.render(...): the reference's shape -- tick hands the index list back
//           to JavaScript (index.js:201-207), 3 MB per frame at 1 M splats
//   queued  comp.frameQueued(camera, viewport) ... comp.sync()    frames queued on the pipeline lanes, each copied behind its
//           kernels into one of `queueDepth` page-locked frames; sync() every queueDepth frames
// Prints ONE JSON line.
'use strict';
const { GaussianSplatting } = require('./gaussian_splatting.js');
const [scenePath, Ws, Hs, framesS, depthS] = process.argv.slice(2);
const W = Number(Ws), H = Number(Hs), frames = Number(framesS || 240), depth = Number(depthS || 48);
function composeYaw(p, yawDeg) {                     // three.js Matrix4.compose: translation p, rotation about +y, unit scale
  const h = yawDeg * Math.PI / 360, y = Math.sin(h), w = Math.cos(h), y2 = y + y, yy = y * y2, wy = w * y2;
  return { elements: [1 - yy, 0, -wy, 0, 0, 1, 0, 0, wy, 0, 1 - yy, 0, p[0], p[1], p[2], 1] };
}
function perspective(fovDeg, aspect, near, far) {    // three.js PerspectiveCamera.updateProjectionMatrix
  const top = near * Math.tan(fovDeg * Math.PI / 360), height = 2 * top, width = aspect * height;
  return { elements: [2 * near / width, 0, 0, 0, 0, 2 * near / height, 0, 0, 0, 0, -(far + near) / (far - near), -1, 0, 0, -2 * far * near / (far - near), 0] };
}
const camera = { matrixWorld: composeYaw([0, 1.6, 0], 0), projectionMatrix: perspective(80, W / H, 0.005, 10000) };   // index.html:13
const object = { matrixWorld: composeYaw([0, 1.5, -2], 0) };
const comp = new GaussianSplatting({ src: scenePath }).init(null);
GaussianSplatting.QUEUE_DEPTH = depth;
const vp = { width: W, height: H };
function pose(i) { object.matrixWorld = composeYaw([0, 1.5, -2], 3 * (i % 120)); }
function trySync() { try { comp.sync(); return 0; } catch (e) { if (e.code !== 'GS-9') throw e; return 1; } }
comp.loadData(camera, object, null, scenePath).then((n) => {
  for (let i = 0; i < 120; i++) { pose(i); comp.tick(); comp.render(camera, vp); }              // buffers sized, share settled
  let t0 = process.hrtime.bigint(), sum = 0;
  for (let i = 0; i < frames; i++) { pose(i); comp.tick(); const img = comp.render(camera, vp); sum += img[(i * 4099) % img.length]; }
  const tickSec = Number(process.hrtime.bigint() - t0) / 1e9;
  for (let i = 0; i < 24; i++) { pose(i); comp.frame(camera, vp); }
  t0 = process.hrtime.bigint();
  for (let i = 0; i < frames; i++) { pose(i); const img = comp.frame(camera, vp); sum += img[(i * 4099) % img.length]; }
  const syncSec = Number(process.hrtime.bigint() - t0) / 1e9;
  let retries = 0;
  for (let i = 0; i < 2 * depth; i++) { pose(i); comp.frameQueued(camera, vp); if (i % depth === depth - 1) retries += trySync(); }
  retries += trySync();
  retries = 0;
  t0 = process.hrtime.bigint();
  for (let i = 0; i < frames; i++) { pose(i); const f = comp.frameQueued(camera, vp); if (i % depth === depth - 1) { retries += trySync(); sum += f[i % f.length]; } }
  retries += trySync();
  const qSec = Number(process.hrtime.bigint() - t0) / 1e9;
  // the reference's own rhythm (index.js:201-207, 438-455): every animation frame posts a sort if none is in flight (tickAsync) and draws
  // with the last completed order; the event loop gets a turn per frame (setImmediate), which is where the reply is collected
  return (async () => {
    let sorts = 0;
    for (let i = 0; i < 24; i++) { pose(i); const p = comp.tickAsync(); if (p) p.then(() => {}); comp.render(camera, vp); await new Promise(setImmediate); }
    comp.tickFinish();
    t0 = process.hrtime.bigint();
    for (let i = 0; i < frames; i++) {
      pose(i);
      const p = comp.tickAsync(); if (p) p.then(() => { sorts++; });
      const img = comp.render(camera, vp); sum += img[(i * 4099) % img.length];
      await new Promise(setImmediate);
    }
    comp.tickFinish();
    const rhythmSec = Number(process.hrtime.bigint() - t0) / 1e9;
  const st = comp.stats();
  console.log(JSON.stringify({ splats: n, width: W, height: H, frames, queue_depth: depth, node: process.version,
    fps_sync: +(frames / syncSec).toFixed(1), ms_per_frame_sync: +(syncSec / frames * 1e3).toFixed(4),
    fps_tick_render: +(frames / tickSec).toFixed(1), ms_per_frame_tick_render: +(tickSec / frames * 1e3).toFixed(4),
    fps_queued: +(frames / qSec).toFixed(1), queued_GBps: +(frames / qSec * W * H * 4 / 1e9).toFixed(2),
    fps_posted_sorts: +(frames / rhythmSec).toFixed(1), posted_sorts_completed: sorts,
    sync_retries: retries, frames_redrawn_by_sync: st.retriedFrames !== undefined ? st.retriedFrames : (st.retried_frames || 0), checksum: sum }));
  comp.remove();
  })();
}).catch((e) => { console.error('FAIL:', e && e.stack || e); process.exit(1); });
