// HectorSLAM.Main.HectorSLAMProcessor over the library's own processor (reference: HectorSLAM/Main/HectorSLAMProcessor.cs:17-160):
// match against the pyramid, then redraw the maps only if the robot moved or turned enough.  Update is ONE native call
// (slamhip_hsproc_update): the library matches, evaluates the gate of :107-109 on the device with the float operations the reference
// applies on the host, and enqueues the grid update behind the match before the pose is back -- one blocking wait per scan (round 5
// measured 56 us per Update that way against two blocking calls, match 37.5 us + update, from managed code).  MapRep is the processor's
// own pyramid (slamhip_hsproc_hs), so everything a caller reads from it (cells, bitmaps, extends) is what Update wrote.
using System;
using System.Drawing;
using System.Numerics;
using System.Runtime.InteropServices;
using BaseSLAM;
using Microsoft.Extensions.Logging;
using SlamHip;

namespace HectorSLAM.Main
{
    public class HectorSLAMProcessor : IDisposable
    {
        private readonly ILogger logger;
        private readonly Device device;
        private readonly bool ownsDevice;
        private readonly Handle proc;
        private float minDistanceDiff = 0.3f, minAngleDiff = 0.13f;

        public MapRepMultiMap MapRep { get; private set; }
        public Vector3 LastMapUpdatePose { get; private set; }
        public Vector3 MatchPose { get; private set; }
        /// <summary>Moving average of the matching time, ms (HectorSLAMProcessor.cs:41,96; kept by the library).</summary>
        public float MatchTiming { get; private set; }
        /// <summary>Moving average of the map update time, ms (:46,115): the update is enqueued, not waited for, so this is the enqueue.</summary>
        public float UpdateTiming { get; private set; }

        public float MinDistanceDiffForMapUpdate                          // :51
        {
            get => minDistanceDiff;
            set { minDistanceDiff = value; Native.Check(Native.slamhip_hsproc_set_thresholds(proc.Ptr, minDistanceDiff, minAngleDiff)); }
        }

        public float MinAngleDiffForMapUpdate                             // :56
        {
            get => minAngleDiff;
            set { minAngleDiff = value; Native.Check(Native.slamhip_hsproc_set_thresholds(proc.Ptr, minDistanceDiff, minAngleDiff)); }
        }

        public HectorSLAMProcessor(float mapResolution, Point mapSize, Vector3 startPose, int numDepth, int numThreads, ILogger logger = null)
            : this(mapResolution, mapSize, startPose, numDepth, numThreads, logger, null)
        {
        }

        /// <param name="numThreads">the reference's matcher threads (:75); the device needs none</param>
        public HectorSLAMProcessor(float mapResolution, Point mapSize, Vector3 startPose, int numDepth, int numThreads, ILogger logger, Device device)
        {
            this.logger = logger;
            this.device = device ?? new Device(0);
            ownsDevice = device == null;
            Native.Check(Native.slamhip_hsproc_create(this.device.Ctx.Ptr, mapResolution, mapSize.X, mapSize.Y, startPose, numDepth, out IntPtr h));
            proc = new Handle(h, Native.slamhip_hsproc_destroy);
            Native.Check(Native.slamhip_hsproc_hs(proc.Ptr, out IntPtr pyramid));
            MapRep = new MapRepMultiMap(this.device, pyramid, numDepth);
            Refresh();
        }

        /// <returns>true if the maps were redrawn (HectorSLAMProcessor.cs:86-126)</returns>
        public unsafe bool Update(ScanCloud scan, Vector3 poseHintWorld, bool mapWithoutMatching = false)
        {
            int updated;
            fixed (Vector2* p = CollectionsMarshal.AsSpan(scan.Points))
                Native.Check(Native.slamhip_hsproc_update(proc.Ptr, p, scan.Points.Count, new Vector2(scan.Pose.X, scan.Pose.Y), poseHintWorld,
                                                          mapWithoutMatching ? 1 : 0, out updated));
            Refresh();
            if (updated == 0) return false;
            MapRep.MarkStale();
            logger?.LogInformation($"Map update at {MatchPose.X:F3} {MatchPose.Y:F3} {MatchPose.Z:F4}");
            return true;
        }

        private void Refresh()
        {
            Native.Check(Native.slamhip_hsproc_get(proc.Ptr, out Vector3 match, out Vector3 last, out float tm, out float tu));
            MatchPose = match; LastMapUpdatePose = last; MatchTiming = tm; UpdateTiming = tu;
        }

        public void Reset()                                              // :131-138
        {
            Native.Check(Native.slamhip_hsproc_reset(proc.Ptr));
            MapRep.MarkStale();
            Refresh();
        }

        public void Dispose()
        {
            MapRep.Dispose();                                            // (borrows the pyramid: nothing native is released here)
            proc.Dispose();
            if (ownsDevice) device.Dispose();
            GC.SuppressFinalize(this);
        }
    }
}
