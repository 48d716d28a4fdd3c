// CoreSLAM.CoreSLAMProcessor with the hot path on an MI355X: same public members as the reference class
// (CoreSLAM/CoreSLAMProcessor.cs:18-773), same Update state machine, but the Monte-Carlo search (:624-710), the distance
// function (:215-259) and both raster updates (:496-593) are the HIP kernels of libslamhip.so.  What stays managed is what
// the reference does per scan outside those loops: odometry bookkeeping and the polar -> cartesian conversion
// (ScanSegmentsToCloud, with MathF.Cos/Sin exactly as in the reference).
//
// Differences a caller can observe:
//  * the candidate jitters come from the library's seeded Philox generator (slamhip_cs_generate_offsets) instead of
//    Redzen's entropy-seeded Ziggurat sampler (:136-137) -- the reference's stream is not reproducible either; set Seed
//    for repeatable runs, or hand in your own list with SetOffsets (flat thread-major order, X, Y, theta per jitter);
//  * HoleMap.Pixels / ObstacleMap.Pixels are mirrors of the device maps, brought up to date when they are READ (MirrorMode.OnRead,
//    the default), by a request after every Update (EveryScan) or only on demand (Manual);
//  * NumSearchThreads only sizes the candidate list (threads x iterations, as in :674-710); there is no thread pool.
using System;
using System.Collections.Generic;
using System.Numerics;
using System.Runtime.InteropServices;
using BaseSLAM;
using SlamHip;

namespace CoreSLAM
{
    /// <summary>When HoleMap.Pixels / ObstacleMap.Pixels follow the device maps (CoreSLAMProcessor.MirrorMode).</summary>
    public enum MirrorMode { OnRead, EveryScan, Manual }

    public class CoreSLAMProcessor : IDisposable
    {
        private readonly Device device;
        private readonly bool ownsDevice;
        private readonly Handle cs;
        private readonly Vector3 startPose;
        private readonly List<Vector2> cloud = new List<Vector2>();
        private Vector3 lastOdometryPose;
        private int scanCount;
        private ulong scanNumber;
        private bool pinnedOffsets;

        public float PhysicalMapSize { get; }
        public HoleMap HoleMap { get; }
        public ObstacleMap ObstacleMap { get; }
        public float SigmaXY { get; }
        public float SigmaTheta { get; }
        public int SearchIterationsPerThread { get; }
        public int NumSearchThreads { get; }

        /// <summary>Blend weight of a new measurement, 0..255 (CoreSLAMProcessor.cs:80).</summary>
        public byte Quality { get; set; } = 50;
        /// <summary>Width of the hole drawn around an obstacle, metres (:85).</summary>
        public float HoleWidth { get; set; } = 0.6f;
        /// <summary>Scans taken on trust (odometry pose, no search) before the search begins (:90).</summary>
        public int PositionSearchBeginning { get; set; } = 5;
        /// <summary>ObstacleMap reset value (:96).</summary>
        public sbyte UnmappedObstacleHits { get; set; } = -5;
        /// <summary>ObstacleMap saturation (:101).</summary>
        public sbyte MaxObstacleHits { get; set; } = 10;
        /// <summary>Last estimated pose: X, Y in metres, Z in radians (:106).</summary>
        public Vector3 Pose { get; private set; } = Vector3.Zero;

        /// <summary>Seed of the candidate generator (new: the reference seeds from entropy).</summary>
        public ulong Seed { get; set; } = 0x5EED5EEDUL;
        /// <summary>Opt-in (new): the candidates' headings on a lattice -- the candidates one lane of the search kernel evaluates share
        /// their heading and differ in translation (slamhip_cs_generate_offsets_lattice): 5 - 7 % faster searches from 65 536
        /// candidates on, the same localisation quality on the simulator's lap.  Off: every candidate has its own (stratified)
        /// heading, as close to the reference's independent draws as a reproducible generator gets.</summary>
        public bool UseHeadingLattice { get; set; } = false;
        /// <summary>When the managed map mirrors (HoleMap.Pixels, ObstacleMap.Pixels) are brought up to date.
        /// OnRead (default): an Update only marks them stale; the `Pixels` getters fetch what changed since their last read --
        /// the HoleMap through slamhip_cs_holemap_mirror_async + _wait (the 16-byte units that changed, ~0.25 ms at 2048 x 2048 for
        /// a display that reads every few dozen scans), the small ObstacleMap whole -- so a caller that does not look at the maps
        /// pays nothing and a reader always sees the current map (bench.py: us_per_scan_with_mirror_read_every_33_scans).
        /// EveryScan: additionally every Update issues the HoleMap's asynchronous request, so that a read right after a scan finds
        /// the push already under way (a request per scan throttles a 20 k scans/s loop: the push is PCIe-store bound).
        /// Manual: only Download() / Mirror() / MirrorAsync() refresh the arrays.</summary>
        public MirrorMode MirrorMode
        {
            get => mirrorMode;
            set { mirrorMode = value; HoleMap.Mode = value; ObstacleMap.Mode = value; }
        }
        private MirrorMode mirrorMode = MirrorMode.OnRead;
        /// <summary>Source compatibility with the earlier shim: false = MirrorMode.Manual, true = OnRead.</summary>
        public bool MirrorMaps
        {
            get => mirrorMode != MirrorMode.Manual;
            set => MirrorMode = value ? MirrorMode.OnRead : MirrorMode.Manual;
        }

        public CoreSLAMProcessor(float physicalMapSize, int holeMapSize, int obstacleMapSize, Vector3 startPose,
                                 float sigmaXY, float sigmaTheta, int iterationsPerThread, int numSearchThreads)
            : this(physicalMapSize, holeMapSize, obstacleMapSize, startPose, sigmaXY, sigmaTheta, iterationsPerThread, numSearchThreads, null)
        {
        }

        /// <param name="device">GPU to run on; null = a private Device(0).</param>
        public CoreSLAMProcessor(float physicalMapSize, int holeMapSize, int obstacleMapSize, Vector3 startPose,
                                 float sigmaXY, float sigmaTheta, int iterationsPerThread, int numSearchThreads, Device device)
        {
            this.device = device ?? new Device(0);
            ownsDevice = device == null;
            this.startPose = startPose;
            PhysicalMapSize = physicalMapSize;
            SigmaXY = sigmaXY;
            SigmaTheta = sigmaTheta;
            SearchIterationsPerThread = iterationsPerThread;
            NumSearchThreads = numSearchThreads;

            Native.Check(Native.slamhip_cs_create(this.device.Ctx.Ptr, physicalMapSize, holeMapSize, obstacleMapSize, out IntPtr h));
            cs = new Handle(h, Native.slamhip_cs_destroy);
            Native.Check(Native.slamhip_cs_info(cs.Ptr, out int hs, out float hscale, out int os, out float oscale));
            HoleMap = new HoleMap(cs, hs, hscale);
            ObstacleMap = new ObstacleMap(cs, os, oscale);
            Reset();
        }

        /// <summary>Maps to their initial state, pose to the start pose (CoreSLAMProcessor.cs:167-175).</summary>
        public void Reset()
        {
            Native.Check(Native.slamhip_cs_reset(cs.Ptr, UnmappedObstacleHits));
            Pose = startPose;
            lastOdometryPose = Vector3.Zero;
            scanCount = 0;
            MapsChanged();
        }

        /// <summary>Use this jitter list (n x (dx, dy, dtheta), flat thread-major order) instead of generated ones.</summary>
        public unsafe void SetOffsets(ReadOnlySpan<Vector3> offsets)
        {
            fixed (Vector3* p = offsets)
                Native.Check(Native.slamhip_cs_set_offsets(cs.Ptr, p, offsets.Length));
            pinnedOffsets = true;
        }

        /// <summary>One revolution of the lidar, possibly delivered in segments with their own odometry poses
        /// (CoreSLAMProcessor.cs:717-752).</summary>
        public unsafe void Update(List<ScanSegment> segments)
        {
            Vector3 odometry = segments[segments.Count - 1].Pose;                  // :719
            SegmentsToCloud(segments, odometry);                                    // :723 (:187-207)
            fixed (Vector2* p = CollectionsMarshal.AsSpan(cloud))
                Native.Check(Native.slamhip_cs_set_scan(cs.Ptr, p, cloud.Count));

            if (scanCount >= PositionSearchBeginning && cloud.Count > 0)            // :726
            {
                Vector3 search = Pose + (odometry - lastOdometryPose);              // :728
                if (!pinnedOffsets)
                {
                    int n = Math.Max(NumSearchThreads, 1) * SearchIterationsPerThread;
                    // (asked for with the scan number as the stream: the library prepares the list of scanNumber + 1 ahead, under
                    // this scan's search, and the next call finds it in place)
                    if (UseHeadingLattice) Native.Check(Native.slamhip_cs_generate_offsets_lattice(cs.Ptr, n, SigmaXY, SigmaTheta, Seed, scanNumber));
                    else Native.Check(Native.slamhip_cs_generate_offsets(cs.Ptr, n, SigmaXY, SigmaTheta, Seed, scanNumber));
                }
                scanNumber++;
                lastOdometryPose = odometry;                                        // :745
                // search (:732), NormalizeAngle (:746) and both map updates (:750-751): one call.  It returns when the pose is
                // on the host; the map updates are enqueued behind the search and finish ~40 us later -- every later call that
                // touches the maps (the next search, the mirrors' downloads below) is ordered behind them on the device
                Native.Check(Native.slamhip_cs_search_and_update(cs.Ptr, search, HoleWidth, Quality, MaxObstacleHits,
                                                                 out Vector3 found, out _, out _));
                Pose = found;
            }
            else
            {
                Vector3 p3;
                if (scanCount < PositionSearchBeginning)
                {
                    scanCount++;                                                    // :741
                    p3 = odometry;                                                  // :742
                }
                else
                {
                    p3 = Pose + (odometry - lastOdometryPose);                      // empty cloud: the un-jittered pose wins (:257, :626-628)
                }
                lastOdometryPose = odometry;
                p3.Z = MathEx.NormalizeAngle(p3.Z);                                 // :746
                Pose = p3;
                if (cloud.Count > 0)
                {
                    Native.Check(Native.slamhip_cs_update_holemap(cs.Ptr, Pose, HoleWidth, Quality));          // :750
                    Native.Check(Native.slamhip_cs_update_obstaclemap(cs.Ptr, Pose, MaxObstacleHits));         // :751
                }
            }
            MapsChanged();
        }

        // the maps on the device moved on: the mirrors are stale (and, in EveryScan mode, the HoleMap's push starts now)
        private void MapsChanged()
        {
            HoleMap.MarkStale();
            ObstacleMap.MarkStale();
            if (mirrorMode == MirrorMode.EveryScan) HoleMap.MirrorAsync();
        }

        // ScanSegmentsToCloud (:187-207): every segment's rays in the frame of the last odometry pose.
        private void SegmentsToCloud(List<ScanSegment> segments, Vector3 odometry)
        {
            cloud.Clear();
            foreach (ScanSegment segment in segments)
            {
                Vector3 rel = segment.Pose - odometry;
                foreach (Ray ray in segment.Rays)
                {
                    float a = ray.Angle + rel.Z;
                    cloud.Add(new Vector2(rel.X + ray.Radius * MathF.Cos(a), rel.Y + ray.Radius * MathF.Sin(a)));
                }
            }
        }

        public void Dispose()
        {
            Dispose(true);
            GC.SuppressFinalize(this);
        }

        protected virtual void Dispose(bool disposing)
        {
            if (!disposing) return;
            cs.Dispose();
            if (ownsDevice) device.Dispose();
        }
    }
}
