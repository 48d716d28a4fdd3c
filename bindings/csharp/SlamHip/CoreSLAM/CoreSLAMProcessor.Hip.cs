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
//  * NumSearchThreads only sizes the candidate list (threads x iterations, as in :674-710); there is no thread pool;
//  * TrigMode.Device (default): (c, s) of every pose are the library's deterministic cos / sin -- the correctly rounded float in all
//    sampled cases, the same on every platform, but not bit-for-bit the host's MathF (0.04 % of the distances of a search differ
//    from a glibc-CRT run, the arg-min moved in none of 2000 searches).  TrigMode.Host: the shim forms px, py, c, s of every
//    candidate with MathF exactly as CalculateDistanceSISD (:232-235) and the two map updates (:499-502, :545-548) do, and the
//    library only gathers, sums and draws: distances, winner and both maps are then IDENTICAL to the reference on this host (at the
//    price of the candidates crossing PCIe and 16 385 MathF.Cos / Sin pairs per scan on the CPU);
//  * a constructor that takes several device ordinals runs the search on all of them (candidate blocks, one 8-byte RCCL min
//    all-reduce per scan, the map updates as bit-identical replicas: slamhip_group_*).
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

    /// <summary>Who forms cos / sin of the poses: the library's deterministic routine, or the host's MathF (see the file header).</summary>
    public enum TrigMode { Device, Host }

    public class CoreSLAMProcessor : IDisposable
    {
        private readonly Device device;
        private readonly bool ownsDevice;
        private readonly Handle cs;
        private readonly Handle group;                                  // several GPUs: the slamhip_group that owns `cs` (its rank 0) and the others
        private Vector3[] hostOffsets = Array.Empty<Vector3>();         // TrigMode.Host: the jitters of the scan on the host
        private Vector4[] hostCandidates = Array.Empty<Vector4>();
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

        /// <summary>Device (default): poses go to the library, which forms cos / sin itself.  Host: MathF on this side of the boundary,
        /// results identical to the reference on this host (single GPU only).</summary>
        public TrigMode TrigMode { get; set; } = TrigMode.Device;
        /// <summary>GPUs the search runs on (1 unless the multi-GPU constructor was used).</summary>
        public int DeviceCount { get; } = 1;

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
            : this(physicalMapSize, holeMapSize, obstacleMapSize, startPose, sigmaXY, sigmaTheta, iterationsPerThread, numSearchThreads, (Device)null)
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

        /// <summary>The search on several GPUs of this host (one process, RCCL over xGMI): every GPU holds the maps and the scan and
        /// evaluates a contiguous block of the candidate list; one 8-byte min all-reduce per scan replaces the cross-thread arg-min
        /// of CoreSLAMProcessor.cs:695-705; the map updates run as replicas (bit-exact kernels keep them identical --
        /// ReplicasEqual() checks).  HoleMap / ObstacleMap mirror the first GPU's maps.</summary>
        public unsafe CoreSLAMProcessor(float physicalMapSize, int holeMapSize, int obstacleMapSize, Vector3 startPose,
                                        float sigmaXY, float sigmaTheta, int iterationsPerThread, int numSearchThreads, int[] deviceOrdinals)
        {
            if (deviceOrdinals == null || deviceOrdinals.Length == 0) throw new ArgumentException("at least one device ordinal", nameof(deviceOrdinals));
            this.startPose = startPose;
            PhysicalMapSize = physicalMapSize;
            SigmaXY = sigmaXY;
            SigmaTheta = sigmaTheta;
            SearchIterationsPerThread = iterationsPerThread;
            NumSearchThreads = numSearchThreads;
            IntPtr g;
            fixed (int* d = deviceOrdinals)
                Native.Check(Native.slamhip_group_create(d, deviceOrdinals.Length, physicalMapSize, holeMapSize, obstacleMapSize, out g));
            group = new Handle(g, Native.slamhip_group_destroy);
            Native.Check(Native.slamhip_group_size(group.Ptr, out int n));
            DeviceCount = n;
            Native.Check(Native.slamhip_group_cs(group.Ptr, 0, out IntPtr h0));
            cs = new Handle(h0, _ => 0);                               // (owned by the group: destroyed with it)
            Native.Check(Native.slamhip_cs_info(cs.Ptr, out int hs, out float hscale, out int os, out float oscale));
            HoleMap = new HoleMap(cs, hs, hscale);
            ObstacleMap = new ObstacleMap(cs, os, oscale);
            Reset();
        }

        /// <summary>Several GPUs: do all replicas hold the same two maps (checksums compared on the host)?</summary>
        public bool ReplicasEqual()
        {
            if (group == null) return true;
            Native.Check(Native.slamhip_group_replicas_equal(group.Ptr, out int eq));
            return eq != 0;
        }

        /// <summary>Maps to their initial state, pose to the start pose (CoreSLAMProcessor.cs:167-175).</summary>
        public void Reset()
        {
            if (group != null) Native.Check(Native.slamhip_group_reset(group.Ptr, UnmappedObstacleHits));
            else Native.Check(Native.slamhip_cs_reset(cs.Ptr, UnmappedObstacleHits));
            Pose = startPose;
            lastOdometryPose = Vector3.Zero;
            scanCount = 0;
            MapsChanged();
        }

        /// <summary>Use this jitter list (n x (dx, dy, dtheta), flat thread-major order) instead of generated ones.</summary>
        public unsafe void SetOffsets(ReadOnlySpan<Vector3> offsets)
        {
            fixed (Vector3* p = offsets)
            {
                if (group != null) Native.Check(Native.slamhip_group_set_offsets(group.Ptr, p, offsets.Length));
                else Native.Check(Native.slamhip_cs_set_offsets(cs.Ptr, p, offsets.Length));
            }
            hostOffsets = offsets.ToArray();
            pinnedOffsets = true;
        }

        // px, py, c, s of a pose at a map scale, with the host's MathF (CoreSLAMProcessor.cs:232-235; :499-502; :545-548)
        private static Vector4 Pxcs(Vector3 pose, float scale) =>
            new Vector4(pose.X * scale + 0.5f, pose.Y * scale + 0.5f, MathF.Cos(pose.Z) * scale, MathF.Sin(pose.Z) * scale);

        // TrigMode.Host: the scan's search and updates with this side's trigonometry (slamhip.h: slamhip_cs_distance_pxcs +
        // slamhip_cs_update_maps_pxcs -- the update rows are those of the NORMALISED winner, as :746 precedes :750-751)
        private unsafe Vector3 SearchAndUpdateHostTrig(Vector3 search)
        {
            int n = pinnedOffsets ? hostOffsets.Length : Math.Max(NumSearchThreads, 1) * SearchIterationsPerThread;
            if (!pinnedOffsets)
            {
                if (hostOffsets.Length != n) hostOffsets = new Vector3[n];
                fixed (Vector3* po = hostOffsets)
                    Native.Check(Native.slamhip_cs_offsets_download(cs.Ptr, po, n));
            }
            if (hostCandidates.Length != n + 1) hostCandidates = new Vector4[n + 1];
            hostCandidates[0] = Pxcs(search, HoleMap.Scale);                                   // :626-628
            for (int k = 0; k < n; k++) hostCandidates[k + 1] = Pxcs(search + hostOffsets[k], HoleMap.Scale);   // :635-637
            int best;
            fixed (Vector4* pc = hostCandidates)
                Native.Check(Native.slamhip_cs_distance_pxcs(cs.Ptr, pc, n + 1, null, out best, out _));
            Vector3 found = best == 0 ? search : search + hostOffsets[best - 1];
            found.Z = MathEx.NormalizeAngle(found.Z);                                           // :746
            Native.Check(Native.slamhip_cs_update_maps_pxcs(cs.Ptr, Pxcs(found, HoleMap.Scale), Pxcs(found, ObstacleMap.Scale),
                                                            HoleWidth, Quality, MaxObstacleHits));   // :750-751
            return found;
        }

        /// <summary>One revolution of the lidar, possibly delivered in segments with their own odometry poses
        /// (CoreSLAMProcessor.cs:717-752).</summary>
        public unsafe void Update(List<ScanSegment> segments)
        {
            Vector3 odometry = segments[segments.Count - 1].Pose;                  // :719
            SegmentsToCloud(segments, odometry);                                    // :723 (:187-207)
            if (group != null && TrigMode == TrigMode.Host) throw new NotSupportedException("TrigMode.Host runs on one GPU");
            bool searching = scanCount >= PositionSearchBeginning && cloud.Count > 0;   // :726
            // One GPU, the library's trigonometry: set_scan (:723), the search (:732) and the updates (:746-751) are ONE call below --
            // slamhip_cs_scan_search_and_update, which may put the search launch into the stream before the scan's tables are made
            // (the candidates are generated first: they do not depend on the scan).  Every other path sets the scan here.
            bool oneCall = searching && group == null && TrigMode != TrigMode.Host;
            if (!oneCall)
                fixed (Vector2* p = CollectionsMarshal.AsSpan(cloud))
                {
                    if (group != null) Native.Check(Native.slamhip_group_set_scan(group.Ptr, p, cloud.Count));
                    else Native.Check(Native.slamhip_cs_set_scan(cs.Ptr, p, cloud.Count));
                }

            if (searching)
            {
                Vector3 search = Pose + (odometry - lastOdometryPose);              // :728
                if (!pinnedOffsets)
                {
                    int n = Math.Max(NumSearchThreads, 1) * SearchIterationsPerThread;
                    // (asked for with the scan number as the stream: the library prepares the list of scanNumber + 1 ahead, under
                    // this scan's search, and the next call finds it in place)
                    if (group != null) Native.Check(Native.slamhip_group_generate_offsets(group.Ptr, n, SigmaXY, SigmaTheta, Seed, scanNumber));
                    else if (UseHeadingLattice) Native.Check(Native.slamhip_cs_generate_offsets_lattice(cs.Ptr, n, SigmaXY, SigmaTheta, Seed, scanNumber));
                    else Native.Check(Native.slamhip_cs_generate_offsets(cs.Ptr, n, SigmaXY, SigmaTheta, Seed, scanNumber));
                }
                scanNumber++;
                lastOdometryPose = odometry;                                        // :745
                if (TrigMode == TrigMode.Host)
                {
                    Pose = SearchAndUpdateHostTrig(search);
                    MapsChanged();
                    return;
                }
                if (group != null)
                {
                    // every GPU: its block of the candidates, the 8-byte min all-reduce, the winner decoded on the device, the replica's updates
                    Native.Check(Native.slamhip_group_search_and_update(group.Ptr, search, HoleWidth, Quality, MaxObstacleHits,
                                                                         out Vector3 foundG, out _, out _));
                    Pose = foundG;
                    MapsChanged();
                    return;
                }
                // the scan (:723), search (:732), NormalizeAngle (:746) and both map updates (:750-751): one call.  It returns when the
                // pose is on the host; the map updates are enqueued behind the search and finish ~40 us later -- every later call that
                // touches the maps (the next search, the mirrors' downloads below) is ordered behind them on the device
                Vector3 found;
                fixed (Vector2* p = CollectionsMarshal.AsSpan(cloud))
                    Native.Check(Native.slamhip_cs_scan_search_and_update(cs.Ptr, p, cloud.Count, search, HoleWidth, Quality, MaxObstacleHits,
                                                                          out found, out _, out _));
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
                    if (group != null)
                        Native.Check(Native.slamhip_group_update_maps(group.Ptr, Pose, HoleWidth, Quality, MaxObstacleHits));       // :750-751 on every replica
                    else if (TrigMode == TrigMode.Host)
                        Native.Check(Native.slamhip_cs_update_maps_pxcs(cs.Ptr, Pxcs(Pose, HoleMap.Scale), Pxcs(Pose, ObstacleMap.Scale),
                                                                        HoleWidth, Quality, MaxObstacleHits));
                    else
                    {
                        Native.Check(Native.slamhip_cs_update_holemap(cs.Ptr, Pose, HoleWidth, Quality));          // :750
                        Native.Check(Native.slamhip_cs_update_obstaclemap(cs.Ptr, Pose, MaxObstacleHits));         // :751
                    }
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
            group?.Dispose();
            if (ownsDevice) device?.Dispose();
        }
    }
}
