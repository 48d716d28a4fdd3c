// HectorSLAM.Main.MapRepMultiMap on the GPU (reference: HectorSLAM/Main/MapRepMultiMap.cs:19-97): level i has
// size / 2^i cells of resolution * 2^i metres; the levels are independent maps drawn from the same scan.  All levels live
// in ONE native pyramid and are updated by one pair of launches (the reference runs a Parallel.ForEach over them, :76).
using System;
using System.Drawing;
using System.Numerics;
using System.Runtime.InteropServices;
using BaseSLAM;
using HectorSLAM.Map;
using SlamHip;

namespace HectorSLAM.Main
{
    public class MapRepMultiMap : IDisposable
    {
        internal readonly Device Device;
        private readonly bool ownsDevice;
        internal readonly Handle Pyramid;

        public int NumLevels => Maps.Length;

        public OccGridMap[] Maps { get; }

        /// <param name="startCoords">must be Vector2.Zero (the only value the reference passes, HectorSLAMProcessor.cs:71)</param>
        public MapRepMultiMap(float mapResolution, Point mapSize, int numDepth, Vector2 startCoords, Device device = null)
        {
            if (startCoords != Vector2.Zero)
                throw new NotSupportedException("the device maps have no offset");
            Device = device ?? new Device(0);
            ownsDevice = device == null;
            Native.Check(Native.slamhip_hs_create(Device.Ctx.Ptr, mapResolution, mapSize.X, mapSize.Y, numDepth, out IntPtr h));
            Pyramid = new Handle(h, Native.slamhip_hs_destroy);
            Maps = new OccGridMap[numDepth];
            for (int i = 0; i < numDepth; i++)
            {
                Maps[i] = new OccGridMap(Device, Pyramid, i);
                Maps[i].IterationsChanged = PushIterations;
                Maps[i].FactorsChanged = () => { };                     // per-level factor setters exist for source compatibility; the pyramid's factors are set below
            }
            PushIterations();
        }

        /// <summary>The pyramid of a native HectorSLAMProcessor (slamhip_hsproc_hs): borrowed, the processor destroys it.</summary>
        internal MapRepMultiMap(Device device, IntPtr borrowedPyramid, int numDepth)
        {
            Device = device;
            ownsDevice = false;
            Pyramid = new Handle(borrowedPyramid, _ => 0);
            Maps = new OccGridMap[numDepth];
            for (int i = 0; i < numDepth; i++)
            {
                Maps[i] = new OccGridMap(Device, Pyramid, i);
                Maps[i].IterationsChanged = PushIterations;
                Maps[i].FactorsChanged = () => { };
            }
            PushIterations();
        }

        private unsafe void PushIterations()
        {
            int* it = stackalloc int[Maps.Length];
            for (int i = 0; i < Maps.Length; i++) it[i] = Maps[i] != null ? Maps[i].EstimateIterations : 3;
            Native.Check(Native.slamhip_hs_set_iterations(Pyramid.Ptr, it));
        }

        public void Reset()                                              // MapRepMultiMap.cs:63-66
        {
            Native.Check(Native.slamhip_hs_reset(Pyramid.Ptr));
            foreach (OccGridMap m in Maps) m.mirrorStale = true;
        }

        /// <summary>Every level from the same scan (MapRepMultiMap.cs:73-77).</summary>
        public unsafe void UpdateByScan(ScanCloud scan, Vector3 pose)
        {
            SetScan(scan);
            Native.Check(Native.slamhip_hs_update_by_scan(Pyramid.Ptr, pose));
            foreach (OccGridMap m in Maps) m.mirrorStale = true;
        }

        /// <summary>The device maps changed behind this object's back (HectorSLAMProcessor.Update drives the native processor): host mirrors are stale.</summary>
        internal void MarkStale()
        {
            foreach (OccGridMap m in Maps) m.mirrorStale = true;
        }

        // The reference reads the scan at call time (ScanMatcher.cs:149-195, OccGridMap.cs:114-239): a caller may refill one
        // ScanCloud in place, or change its Pose, between two calls.  Every call therefore hands the scan to the library again --
        // slamhip_hs_set_scan is a host memcpy into a pinned staging block; the upload itself is deferred to the first kernel that
        // reads the points, so matching and then updating with one scan still uploads it once per call pair at most.
        internal unsafe void SetScan(ScanCloud scan)
        {
            fixed (Vector2* p = CollectionsMarshal.AsSpan(scan.Points))
                Native.Check(Native.slamhip_hs_set_scan(Pyramid.Ptr, p, scan.Points.Count, new Vector2(scan.Pose.X, scan.Pose.Y)));
        }

        public void SetUpdateFactorFree(float factor)                    // MapRepMultiMap.cs:83-89
        {
            foreach (OccGridMap m in Maps) m.SetFactorsSilently(factor, m.UpdateOccupiedFactor);
            Native.Check(Native.slamhip_hs_set_factors(Pyramid.Ptr, factor, Maps[0].UpdateOccupiedFactor));
        }

        public void SetUpdateFactorOccupied(float factor)                // MapRepMultiMap.cs:92-95
        {
            foreach (OccGridMap m in Maps) m.SetFactorsSilently(m.UpdateFreeFactor, factor);
            Native.Check(Native.slamhip_hs_set_factors(Pyramid.Ptr, Maps[0].UpdateFreeFactor, factor));
        }

        public void Dispose()
        {
            Pyramid.Dispose();
            if (ownsDevice) Device.Dispose();
            GC.SuppressFinalize(this);
        }
    }
}
