// HectorSLAM.Map.GridMap / OccGridMap on the GPU.  One native pyramid (slamhip_hs) holds every level of a MapRepMultiMap;
// an OccGridMap object is a VIEW of one level (or, constructed on its own, of a private one-level pyramid).  Public members
// follow the reference classes (HectorSLAM/Map/GridMap.cs:11-209, OccGridMap.cs:11-253); the cells live in device memory as
// {float Value; int UpdateIndex} planes plus a plane of cached probabilities that every writer keeps current.
// LogOddsCell and MapProperties are the reference's own plain data types (HectorSLAM/Map/LogOddsCell.cs, MapProperties.cs),
// used unchanged.
using System;
using System.Drawing;
using System.Numerics;
using System.Runtime.InteropServices;
using BaseSLAM;
using SlamHip;

namespace HectorSLAM.Map
{
    public class GridMap
    {
        internal readonly Device Device;
        internal readonly bool OwnsDevice;
        internal readonly Handle Pyramid;                               // slamhip_hs
        internal readonly bool OwnsPyramid;
        internal readonly int Level;

        /// <summary>Cell length, dimensions, offset (GridMap.cs:20).</summary>
        public MapProperties Properties { get; }

        /// <summary>Map size in cells (GridMap.cs:25).</summary>
        public Point Dimensions => Properties.Dimensions;

        /// <summary>A map of its own: a one-level pyramid on `device` (null: a private Device(0)).</summary>
        public GridMap(float mapResolution, Point size, Vector2 offset, Device device = null)
        {
            if (offset != Vector2.Zero)
                throw new NotSupportedException("the device maps have no offset (the reference always passes Vector2.Zero, HectorSLAMProcessor.cs:71)");
            Device = device ?? new Device(0);
            OwnsDevice = device == null;
            Native.Check(Native.slamhip_hs_create(Device.Ctx.Ptr, mapResolution, size.X, size.Y, 1, out IntPtr h));
            Pyramid = new Handle(h, Native.slamhip_hs_destroy);
            OwnsPyramid = true;
            Level = 0;
            Properties = new MapProperties(mapResolution, size, offset);
        }

        /// <summary>Level `level` of an existing pyramid (MapRepMultiMap).</summary>
        internal GridMap(Device device, Handle pyramid, int level)
        {
            Device = device;
            Pyramid = pyramid;
            Level = level;
            Native.Check(Native.slamhip_hs_level_info(pyramid.Ptr, level, out int w, out int h, out float cell));
            Properties = new MapProperties(cell, new Point(w, h), Vector2.Zero);
        }

        /// <summary>All cells back to "unknown" (GridMap.cs:56-64).  On a level of a shared pyramid this resets every level,
        /// as MapRepMultiMap.Reset does; the reference never resets a single level of a pyramid.</summary>
        public virtual void Reset() => Native.Check(Native.slamhip_hs_reset(Pyramid.Ptr));

        public LogOddsCell GetCell(int x, int y) => GetCell(y * Dimensions.X + x);          // GridMap.cs:71

        public LogOddsCell GetCell(Point point) => GetCell(point.X, point.Y);               // :82

        /// <summary>One cell (GridMap.cs:93).  A whole-level read-back per call would be absurd, so the level is fetched once
        /// into a mirror that stays valid until the next UpdateByScan / Reset; bulk readers should use DownloadCells.</summary>
        public LogOddsCell GetCell(int index)
        {
            if (mirror == null || mirrorStale) { mirror = DownloadCells(); mirrorStale = false; }
            return mirror[index];
        }

        private LogOddsCell[] mirror;
        internal bool mirrorStale = true;

        /// <summary>Every cell of this level, row-major.</summary>
        public unsafe LogOddsCell[] DownloadCells()
        {
            LogOddsCell[] cells = new LogOddsCell[Dimensions.X * Dimensions.Y];
            fixed (LogOddsCell* p = cells)
                Native.Check(Native.slamhip_hs_cells_download(Pyramid.Ptr, Level, p, (nuint)cells.Length));
            return cells;
        }

        /// <summary>Restore a saved level.</summary>
        public unsafe void UploadCells(LogOddsCell[] cells)
        {
            fixed (LogOddsCell* p = cells)
                Native.Check(Native.slamhip_hs_cells_upload(Pyramid.Ptr, Level, p, (nuint)cells.Length));
            mirrorStale = true;
        }

        /// <summary>127 = unscanned, 0 = occupied, 254 = free (GridMap.cs:104-115); produced on the device, one byte per cell
        /// crosses PCIe.</summary>
        public unsafe byte[] GetBitmapData()
        {
            byte[] data = new byte[Dimensions.X * Dimensions.Y];
            fixed (byte* p = data)
                Native.Check(Native.slamhip_hs_bitmap_download(Pyramid.Ptr, Level, p, (nuint)data.Length));
            return data;
        }

        /// <summary>Map pose (cells, radians) to world pose (metres, radians) (GridMap.cs:122-126).</summary>
        public Vector3 GetWorldCoordsPose(Vector3 mapPose) =>
            new Vector3(mapPose.X * Properties.CellLength, mapPose.Y * Properties.CellLength, mapPose.Z);

        /// <summary>World pose to map pose (GridMap.cs:133-137).</summary>
        public Vector3 GetMapCoordsPose(Vector3 worldPose) =>
            new Vector3(worldPose.X * Properties.ScaleToMap, worldPose.Y * Properties.ScaleToMap, worldPose.Z);

        /// <summary>Bounding rectangle of the cells that were ever updated (GridMap.cs:147-207); a device reduction.</summary>
        public unsafe bool GetMapExtends(out int xMax, out int yMax, out int xMin, out int yMin)
        {
            int* e = stackalloc int[4];
            Native.Check(Native.slamhip_hs_map_extends(Pyramid.Ptr, Level, e, out int found));
            xMax = e[0]; yMax = e[1]; xMin = e[2]; yMin = e[3];
            return found != 0;
        }

        internal void DisposeOwned()
        {
            if (OwnsPyramid) Pyramid.Dispose();
            if (OwnsDevice) Device.Dispose();
        }
    }

    public class OccGridMap : GridMap, IDisposable
    {
        private float oddsFree = 0.4f, oddsOccupied = 0.9f;              // OccGridMap.cs:24-27
        internal Action FactorsChanged;                                  // set by MapRepMultiMap: factors are per pyramid on the device

        public OccGridMap(float mapResolution, Point size, Vector2 offset, Device device = null)
            : base(mapResolution, size, offset, device)
        {
            PushFactors();
            PushIterations();
        }

        internal OccGridMap(Device device, Handle pyramid, int level) : base(device, pyramid, level)
        {
        }

        private int estimateIterations = 3;

        /// <summary>Gauss-Newton iterations the matcher spends on this level (OccGridMap.cs:53).</summary>
        public int EstimateIterations
        {
            get => estimateIterations;
            set { estimateIterations = value; IterationsChanged?.Invoke(); if (OwnsPyramid) PushIterations(); }
        }

        internal Action IterationsChanged;

        public float UpdateFreeFactor                                   // OccGridMap.cs:58-66
        {
            get => oddsFree;
            set { oddsFree = value; FactorsChanged?.Invoke(); if (OwnsPyramid) PushFactors(); }
        }

        public float UpdateOccupiedFactor                               // OccGridMap.cs:71-79
        {
            get => oddsOccupied;
            set { oddsOccupied = value; FactorsChanged?.Invoke(); if (OwnsPyramid) PushFactors(); }
        }

        internal void SetFactorsSilently(float free, float occupied) { oddsFree = free; oddsOccupied = occupied; }

        private void PushFactors() => Native.Check(Native.slamhip_hs_set_factors(Pyramid.Ptr, oddsFree, oddsOccupied));

        private unsafe void PushIterations()
        {
            int it = estimateIterations;
            Native.Check(Native.slamhip_hs_set_iterations(Pyramid.Ptr, &it));
        }

        /// <summary>Occupancy probability of one cell (OccGridMap.cs:97-107), read from the device's probability plane.</summary>
        public unsafe float GetCachedProbability(int index)
        {
            float p;
            Native.Check(Native.slamhip_hs_probability(Pyramid.Ptr, Level, &index, 1, &p));
            return p;
        }

        /// <summary>Probabilities of many cells in one call.</summary>
        public unsafe float[] GetCachedProbabilities(int[] indices)
        {
            float[] p = new float[indices.Length];
            fixed (int* i = indices)
            fixed (float* o = p)
                Native.Check(Native.slamhip_hs_probability(Pyramid.Ptr, Level, i, indices.Length, o));
            return p;
        }

        /// <summary>Draw one scan into this map (OccGridMap.cs:114-148).  Only for a map of its own: the levels of a pyramid are
        /// updated together by MapRepMultiMap.UpdateByScan (one launch for all levels).</summary>
        public unsafe void UpdateByScan(ScanCloud scan, Vector3 robotPoseWorld)
        {
            if (!OwnsPyramid)
                throw new InvalidOperationException("this OccGridMap is a level of a MapRepMultiMap: call MapRepMultiMap.UpdateByScan");
            fixed (Vector2* p = CollectionsMarshal.AsSpan(scan.Points))
                Native.Check(Native.slamhip_hs_set_scan(Pyramid.Ptr, p, scan.Points.Count, new Vector2(scan.Pose.X, scan.Pose.Y)));
            Native.Check(Native.slamhip_hs_update_by_scan(Pyramid.Ptr, robotPoseWorld));
            mirrorStale = true;
        }

        public override void Reset()                                    // OccGridMap.cs:244-252
        {
            base.Reset();
            mirrorStale = true;
        }

        public void Dispose()
        {
            DisposeOwned();
            GC.SuppressFinalize(this);
        }
    }
}
