// HectorSLAM.Matcher.ScanMatcher on the GPU (reference: HectorSLAM/Matcher/ScanMatcher.cs:18-272): the Gauss-Newton
// alignment of a scan to the occupancy pyramid -- all levels and iterations of a match are ONE kernel launch (one
// workgroup: bilinear taps, Hessian sums, 3x3 solve, clamp, next iteration).  numThreads is accepted for source
// compatibility; nothing is threaded on the host.  Poses agree with the reference to 1e-4 m / 1e-4 rad (its own result
// moves by that much with its thread count: binary32 chunk sums, :149-195).
using System;
using System.Numerics;
using BaseSLAM;
using HectorSLAM.Main;
using HectorSLAM.Map;
using Microsoft.Extensions.Logging;
using SlamHip;

namespace HectorSLAM.Matcher
{
    public class ScanMatcher : IDisposable
    {
        private readonly ILogger logger;

        public ScanMatcher(int numThreads, ILogger logger = null)
        {
            this.logger = logger;
        }

        /// <summary>Coarse-to-fine over every level of the pyramid (ScanMatcher.cs:41-54).</summary>
        public Vector3 MatchData(MapRepMultiMap multiMap, ScanCloud scan, Vector3 hintPose)
        {
            multiMap.SetScan(scan);
            Native.Check(Native.slamhip_hs_match(multiMap.Pyramid.Ptr, hintPose, out Vector3 pose));
            return pose;
        }

        /// <summary>One grid, gridMap.EstimateIterations iterations (ScanMatcher.cs:64-84).</summary>
        public unsafe Vector3 MatchData(OccGridMap gridMap, ScanCloud scan, Vector3 hintPose)
        {
            if (scan.Points.Count == 0) return hintPose;                 // :82-83
            fixed (Vector2* p = System.Runtime.InteropServices.CollectionsMarshal.AsSpan(scan.Points))
                Native.Check(Native.slamhip_hs_set_scan(gridMap.Pyramid.Ptr, p, scan.Points.Count, new Vector2(scan.Pose.X, scan.Pose.Y)));
            Native.Check(Native.slamhip_hs_match_level(gridMap.Pyramid.Ptr, gridMap.Level, hintPose, gridMap.EstimateIterations, out Vector3 pose));
            return pose;
        }

        /// <summary>Many hints against the same scan and maps in one launch (new: relocalisation, particle filters).</summary>
        public unsafe Vector3[] MatchDataBatch(MapRepMultiMap multiMap, ScanCloud scan, Vector3[] hintPoses)
        {
            multiMap.SetScan(scan);
            Vector3[] poses = new Vector3[hintPoses.Length];
            fixed (Vector3* h = hintPoses)
            fixed (Vector3* o = poses)
                Native.Check(Native.slamhip_hs_match_batch(multiMap.Pyramid.Ptr, h, hintPoses.Length, o));
            return poses;
        }

        public void Dispose()
        {
            GC.SuppressFinalize(this);
        }
    }
}
