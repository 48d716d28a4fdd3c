// HectorSLAM.Main.HectorSLAMProcessor over the GPU matcher and maps (reference: HectorSLAM/Main/HectorSLAMProcessor.cs:17-160):
// match against the pyramid, then redraw the maps only if the robot moved or turned enough.  The state machine is managed
// code as in the reference; the two heavy calls are MatchData and UpdateByScan above.
using System;
using System.Diagnostics;
using System.Drawing;
using System.Numerics;
using BaseSLAM;
using HectorSLAM.Matcher;
using Microsoft.Extensions.Logging;
using SlamHip;

namespace HectorSLAM.Main
{
    public class HectorSLAMProcessor : IDisposable
    {
        private readonly ILogger logger;
        private readonly Vector3 startPose;
        private readonly ScanMatcher scanMatcher;
        private static readonly Vector3 Never = new Vector3(float.MinValue, float.MinValue, float.MinValue);

        public MapRepMultiMap MapRep { get; private set; }
        public Vector3 LastMapUpdatePose { get; private set; }
        public Vector3 MatchPose { get; private set; }
        /// <summary>Moving average of the matching time, ms (HectorSLAMProcessor.cs:41,96).</summary>
        public float MatchTiming { get; private set; }
        /// <summary>Moving average of the map update time, ms (:46,115).</summary>
        public float UpdateTiming { get; private set; }
        public float MinDistanceDiffForMapUpdate { get; set; } = 0.3f;
        public float MinAngleDiffForMapUpdate { get; set; } = 0.13f;

        public HectorSLAMProcessor(float mapResolution, Point mapSize, Vector3 startPose, int numDepth, int numThreads, ILogger logger = null)
            : this(mapResolution, mapSize, startPose, numDepth, numThreads, logger, null)
        {
        }

        public HectorSLAMProcessor(float mapResolution, Point mapSize, Vector3 startPose, int numDepth, int numThreads, ILogger logger, Device device)
        {
            this.logger = logger;
            this.startPose = startPose;
            MapRep = new MapRepMultiMap(mapResolution, mapSize, numDepth, Vector2.Zero, device);
            scanMatcher = new ScanMatcher(numThreads, logger);
            MatchPose = startPose;
            LastMapUpdatePose = Never;
        }

        /// <returns>true if the maps were redrawn (HectorSLAMProcessor.cs:86-126)</returns>
        public bool Update(ScanCloud scan, Vector3 poseHintWorld, bool mapWithoutMatching = false)
        {
            if (mapWithoutMatching)
            {
                MatchPose = poseHintWorld;
            }
            else
            {
                long t0 = Stopwatch.GetTimestamp();
                MatchPose = scanMatcher.MatchData(MapRep, scan, poseHintWorld);
                MatchTiming = (3.0f * MatchTiming + ElapsedMs(t0)) / 4.0f;
            }

            Vector2 moved = new Vector2(MatchPose.X - LastMapUpdatePose.X, MatchPose.Y - LastMapUpdatePose.Y);
            bool farEnough = moved.LengthSquared() > MinDistanceDiffForMapUpdate * MinDistanceDiffForMapUpdate;
            // (the reference compares a difference of RADIANS with DegDiff, a degree wrap -- :108; kept as it is)
            bool turnedEnough = MathEx.DegDiff(MatchPose.Z, LastMapUpdatePose.Z) > MinAngleDiffForMapUpdate;
            if (!(farEnough || turnedEnough || mapWithoutMatching)) return false;

            long t1 = Stopwatch.GetTimestamp();
            MapRep.UpdateByScan(scan, MatchPose);
            UpdateTiming = (3.0f * UpdateTiming + ElapsedMs(t1)) / 4.0f;
            LastMapUpdatePose = MatchPose;
            logger?.LogInformation($"Map update at {MatchPose.X:F3} {MatchPose.Y:F3} {MatchPose.Z:F4}");
            return true;
        }

        private static float ElapsedMs(long since) => (float)((Stopwatch.GetTimestamp() - since) * 1000.0 / Stopwatch.Frequency);

        public void Reset()                                              // :131-138
        {
            MapRep.Reset();
            MatchPose = startPose;
            LastMapUpdatePose = Never;
        }

        public void Dispose()
        {
            scanMatcher.Dispose();
            MapRep.Dispose();
            GC.SuppressFinalize(this);
        }
    }
}
