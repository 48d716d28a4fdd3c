// SlamHip.Native.cs -- P/Invoke surface of libslamhip.so (include/slamhip.h), one declaration per C entry point the
// managed shims use.  Every function returns an int32 status (0 = OK); Check() turns anything else into an
// InvalidOperationException carrying slamhip_last_error().  Structs crossing the boundary are blittable:
// Vector2 / Vector3 / Vector4 (8 / 12 / 16 bytes) and LogOddsCell {int UpdateIndex; float Value} (8 bytes).
//
// Source only: the build image of this repository has no .NET SDK; the same symbols are exercised by the ctypes
// binding (slam.net_amd/capi.py) and by the GPU tests.
using System;
using System.Numerics;
using System.Runtime.InteropServices;

namespace SlamHip
{
    internal static unsafe class Native
    {
        const string Lib = "slamhip";                                   // libslamhip.so on the library path

        [DllImport(Lib)] internal static extern IntPtr slamhip_version();
        [DllImport(Lib)] internal static extern IntPtr slamhip_last_error();
        [DllImport(Lib)] internal static extern int slamhip_device_count(out int count);

        // ---- context: one GPU + one HIP stream; stands where `new ParallelWorker(n)` stood --------------------------
        [DllImport(Lib)] internal static extern int slamhip_ctx_create(int deviceOrdinal, out IntPtr ctx);
        [DllImport(Lib)] internal static extern int slamhip_ctx_destroy(IntPtr ctx);
        [DllImport(Lib)] internal static extern int slamhip_ctx_synchronize(IntPtr ctx);
        [DllImport(Lib)] internal static extern int slamhip_ctx_set_wait_timeout(IntPtr ctx, long timeoutMs);
        [DllImport(Lib)] internal static extern int slamhip_ctx_poisoned(IntPtr ctx, out int poisoned);
        [DllImport(Lib)] internal static extern int slamhip_ctx_philox4x32_10(IntPtr ctx, uint[] counter4, uint[] key2, [Out] uint[] out4);

        // ---- CoreSLAM operator level --------------------------------------------------------------------------------
        [DllImport(Lib)] internal static extern int slamhip_cs_create(IntPtr ctx, float physicalMapSize, int holeMapSize, int obstacleMapSize, out IntPtr cs);
        [DllImport(Lib)] internal static extern int slamhip_cs_destroy(IntPtr cs);
        [DllImport(Lib)] internal static extern int slamhip_cs_info(IntPtr cs, out int holeSize, out float holeScale, out int obstSize, out float obstScale);
        [DllImport(Lib)] internal static extern int slamhip_cs_reset(IntPtr cs, int unmappedObstacleHits);
        [DllImport(Lib)] internal static extern int slamhip_cs_holemap_upload(IntPtr cs, ushort* pixels, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_cs_holemap_download(IntPtr cs, ushort* pixels, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_cs_holemap_download_packed(IntPtr cs, byte* packed, nuint nBytes);
        [DllImport(Lib)] internal static extern int slamhip_cs_holemap_mirror(IntPtr cs, ushort* pixels, nuint n, int* rectX0Y0X1Y1);
        [DllImport(Lib)] internal static extern int slamhip_cs_holemap_mirror_async(IntPtr cs, ushort* pixels, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_cs_holemap_mirror_wait(IntPtr cs, int* rectX0Y0X1Y1, long* pixelsPushed);
        [DllImport(Lib)] internal static extern int slamhip_cs_holemap_mirror_release(IntPtr cs);
        [DllImport(Lib)] internal static extern int slamhip_cs_obstaclemap_upload(IntPtr cs, sbyte* pixels, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_cs_obstaclemap_download(IntPtr cs, sbyte* pixels, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_cs_set_scan(IntPtr cs, Vector2* points, int nPoints);
        [DllImport(Lib)] internal static extern int slamhip_cs_distance_pxcs(IntPtr cs, Vector4* pxcs, int k, int* outDist, out int bestIndex, out int bestDist);
        [DllImport(Lib)] internal static extern int slamhip_cs_distance_poses(IntPtr cs, Vector3* poses, int k, int* outDist, out int bestIndex, out int bestDist);
        [DllImport(Lib)] internal static extern int slamhip_cs_set_offsets(IntPtr cs, Vector3* offs, int n);
        [DllImport(Lib)] internal static extern int slamhip_cs_generate_offsets(IntPtr cs, int n, float sigmaXY, float sigmaTheta, ulong seed, ulong stream);
        [DllImport(Lib)] internal static extern int slamhip_cs_generate_offsets_lattice(IntPtr cs, int n, float sigmaXY, float sigmaTheta, ulong seed, ulong stream);
        [DllImport(Lib)] internal static extern int slamhip_cs_prepared_lists(IntPtr cs, out ulong served, out ulong prepared);
        [DllImport(Lib)] internal static extern int slamhip_cs_prelaunch_stats(IntPtr cs, [Out] ulong[] four);
        [DllImport(Lib)] internal static extern int slamhip_cs_plan_stats(IntPtr cs, [Out] ulong[] four);
        [DllImport(Lib)] internal static extern int slamhip_cs_search(IntPtr cs, in Vector3 searchPose, out Vector3 pose, out int dist, out int index);
        [DllImport(Lib)] internal static extern int slamhip_cs_update_holemap(IntPtr cs, in Vector3 pose, float holeWidth, int quality);
        [DllImport(Lib)] internal static extern int slamhip_cs_update_holemap_pxcs(IntPtr cs, in Vector4 pxcs, float holeWidth, int quality);
        [DllImport(Lib)] internal static extern int slamhip_cs_update_obstaclemap(IntPtr cs, in Vector3 pose, int maxObstacleHits);
        [DllImport(Lib)] internal static extern int slamhip_cs_update_obstaclemap_pxcs(IntPtr cs, in Vector4 pxcs, int maxObstacleHits);
        [DllImport(Lib)] internal static extern int slamhip_cs_search_and_update(IntPtr cs, in Vector3 searchPose, float holeWidth, int quality, int maxObstacleHits,
                                                                                 out Vector3 pose, out int dist, out int index);
        [DllImport(Lib)] internal static extern int slamhip_cs_scan_search_and_update(IntPtr cs, Vector2* xy, int nPoints, in Vector3 searchPose, float holeWidth, int quality, int maxObstacleHits,
                                                                                      out Vector3 pose, out int dist, out int index);
        [DllImport(Lib)] internal static extern int slamhip_cs_search_and_update_pxcs(IntPtr cs, Vector4* pxcsSearch, Vector4* pxcsUpdateHole, Vector4* pxcsUpdateObstacle, int k,
                                                                                      float holeWidth, int quality, int maxObstacleHits, out int index, out int dist);
        [DllImport(Lib)] internal static extern int slamhip_cs_update_maps_pxcs(IntPtr cs, in Vector4 pxcsHole, in Vector4 pxcsObstacle, float holeWidth, int quality, int maxObstacleHits);
        [DllImport(Lib)] internal static extern int slamhip_cs_offsets_download(IntPtr cs, Vector3* offs, int n);

        // ---- HectorSLAM operator level ------------------------------------------------------------------------------
        [DllImport(Lib)] internal static extern int slamhip_hs_create(IntPtr ctx, float cellLength, int width, int height, int levels, out IntPtr hs);
        [DllImport(Lib)] internal static extern int slamhip_hs_destroy(IntPtr hs);
        [DllImport(Lib)] internal static extern int slamhip_hs_reset(IntPtr hs);
        [DllImport(Lib)] internal static extern int slamhip_hs_level_info(IntPtr hs, int level, out int width, out int height, out float cellLength);
        [DllImport(Lib)] internal static extern int slamhip_hs_set_factors(IntPtr hs, float free, float occupied);
        [DllImport(Lib)] internal static extern int slamhip_hs_set_iterations(IntPtr hs, int* perLevel);
        [DllImport(Lib)] internal static extern int slamhip_hs_cells_upload(IntPtr hs, int level, HectorSLAM.Map.LogOddsCell* cells, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_hs_cells_download(IntPtr hs, int level, HectorSLAM.Map.LogOddsCell* cells, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_hs_bitmap_download(IntPtr hs, int level, byte* data, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_hs_map_extends(IntPtr hs, int level, int* xMaxYMaxXMinYMin, out int found);
        [DllImport(Lib)] internal static extern int slamhip_hs_probability(IntPtr hs, int level, int* indices, int n, float* p);
        [DllImport(Lib)] internal static extern int slamhip_hs_set_scan(IntPtr hs, Vector2* points, int nPoints, in Vector2 scanOrigin);
        [DllImport(Lib)] internal static extern int slamhip_hs_match(IntPtr hs, in Vector3 hint, out Vector3 pose);
        [DllImport(Lib)] internal static extern int slamhip_hs_match_level(IntPtr hs, int level, in Vector3 hint, int iterations, out Vector3 pose);
        [DllImport(Lib)] internal static extern int slamhip_hs_match_batch(IntPtr hs, Vector3* hints, int count, Vector3* poses);
        [DllImport(Lib)] internal static extern int slamhip_hs_update_by_scan(IntPtr hs, in Vector3 robotPoseWorld);
        // HectorSLAM, processor level (HectorSLAMProcessor.cs:66-138): the Update state machine in the library -- match, the gate of :107-109 evaluated
        // on the device, the grid update enqueued behind the match before the pose is back (one blocking wait per scan instead of two)
        [DllImport(Lib)] internal static extern int slamhip_hsproc_create(IntPtr ctx, float mapResolution, int width, int height, in Vector3 startPose, int numDepth, out IntPtr proc);
        [DllImport(Lib)] internal static extern int slamhip_hsproc_destroy(IntPtr proc);
        [DllImport(Lib)] internal static extern int slamhip_hsproc_reset(IntPtr proc);
        [DllImport(Lib)] internal static extern int slamhip_hsproc_update(IntPtr proc, Vector2* points, int nPoints, in Vector2 scanOrigin, in Vector3 poseHintWorld, int mapWithoutMatching, out int mapUpdated);
        [DllImport(Lib)] internal static extern int slamhip_hsproc_get(IntPtr proc, out Vector3 matchPose, out Vector3 lastMapUpdatePose, out float matchTimingMs, out float updateTimingMs);
        [DllImport(Lib)] internal static extern int slamhip_hsproc_set_thresholds(IntPtr proc, float minDistanceDiff, float minAngleDiff);
        [DllImport(Lib)] internal static extern int slamhip_hsproc_hs(IntPtr proc, out IntPtr hs);

        // ---- one process, several GPUs -------------------------------------------------------------------------------
        [DllImport(Lib)] internal static extern int slamhip_group_create(int* deviceOrdinals, int n, float physicalMapSize, int holeMapSize, int obstacleMapSize, out IntPtr group);
        [DllImport(Lib)] internal static extern int slamhip_group_destroy(IntPtr group);
        [DllImport(Lib)] internal static extern int slamhip_group_cs(IntPtr group, int rank, out IntPtr cs);
        [DllImport(Lib)] internal static extern int slamhip_group_reset(IntPtr group, int unmappedObstacleHits);
        [DllImport(Lib)] internal static extern int slamhip_group_set_scan(IntPtr group, Vector2* points, int nPoints);
        [DllImport(Lib)] internal static extern int slamhip_group_set_offsets(IntPtr group, Vector3* offs, int n);
        [DllImport(Lib)] internal static extern int slamhip_group_generate_offsets(IntPtr group, int n, float sigmaXY, float sigmaTheta, ulong seed, ulong stream);
        [DllImport(Lib)] internal static extern int slamhip_group_size(IntPtr group, out int n);
        [DllImport(Lib)] internal static extern int slamhip_group_holemap_upload(IntPtr group, ushort* pixels, nuint n);
        [DllImport(Lib)] internal static extern int slamhip_group_search_and_update(IntPtr group, in Vector3 searchPose, float holeWidth, int quality, int maxObstacleHits,
                                                                                    out Vector3 pose, out int dist, out int index);
        [DllImport(Lib)] internal static extern int slamhip_group_search(IntPtr group, in Vector3 searchPose, out Vector3 pose, out int dist, out int index);
        [DllImport(Lib)] internal static extern int slamhip_group_update_maps(IntPtr group, in Vector3 pose, float holeWidth, int quality, int maxObstacleHits);
        [DllImport(Lib)] internal static extern int slamhip_group_replicas_equal(IntPtr group, out int equal);
        [DllImport(Lib)] internal static extern int slamhip_cs_maps_checksum(IntPtr cs, ulong* holeAndObstacleWords);
        [DllImport(Lib)] internal static extern int slamhip_hs_checksum(IntPtr hs, int level, ulong* valueAndUpdateIndexWords);

        internal const int ErrTimeout = -6;                              // SLAMHIP_ERR_TIMEOUT: a blocking wait passed its bound; the context is poisoned

        internal static void Check(int status)
        {
            if (status == ErrTimeout)
                throw new TimeoutException($"slamhip: {Marshal.PtrToStringAnsi(slamhip_last_error())} (the device context is poisoned: dispose the objects on it)");
            if (status != 0)
                throw new InvalidOperationException($"slamhip error {status}: {Marshal.PtrToStringAnsi(slamhip_last_error())}");
        }
    }

    /// <summary>Owner of one native handle (slamhip_ctx / _cs / _hs): Dispose or finalisation calls its *_destroy.</summary>
    internal sealed class Handle : SafeHandle
    {
        private readonly Func<IntPtr, int> destroy;

        internal Handle(IntPtr h, Func<IntPtr, int> destroy) : base(IntPtr.Zero, true)
        {
            SetHandle(h);
            this.destroy = destroy;
        }

        public override bool IsInvalid => handle == IntPtr.Zero;

        internal IntPtr Ptr => handle;

        protected override bool ReleaseHandle() => destroy(handle) == 0;
    }

    /// <summary>One GPU and one HIP stream.  A processor owns one; pass a shared one to put several objects on the same device.</summary>
    public sealed class Device : IDisposable
    {
        internal readonly Handle Ctx;

        public int Ordinal { get; }

        public Device(int ordinal = 0)
        {
            Native.Check(Native.slamhip_ctx_create(ordinal, out IntPtr h));
            Ctx = new Handle(h, Native.slamhip_ctx_destroy);
            Ordinal = ordinal;
        }

        public static int Count
        {
            get { Native.Check(Native.slamhip_device_count(out int n)); return n; }
        }

        public void Synchronize() => Native.Check(Native.slamhip_ctx_synchronize(Ctx.Ptr));

        /// <summary>Bound on every blocking wait on this device, in milliseconds (default 10 000, or SLAMHIP_WAIT_TIMEOUT_MS; 0: none).
        /// The reference's ParallelWorker.Work waits for its threads without a bound; here a kernel that never ends surfaces as a
        /// TimeoutException, after which the device context is poisoned: every later call fails at once, nothing is re-executed, and
        /// the objects on it are to be disposed.</summary>
        public long WaitTimeoutMs
        {
            set => Native.Check(Native.slamhip_ctx_set_wait_timeout(Ctx.Ptr, value));
        }

        /// <summary>True once a blocking wait on this device has timed out.</summary>
        public bool Poisoned
        {
            get { Native.Check(Native.slamhip_ctx_poisoned(Ctx.Ptr, out int p)); return p != 0; }
        }

        public void Dispose() => Ctx.Dispose();
    }
}
