// CoreSLAM.ObstacleMap on the GPU (reference: CoreSLAM/ObstacleMap.cs:10-42): negative = unmapped, 0 = clear,
// positive = hits.  `Pixels` is the managed sbyte[y, x] mirror of the device map.
using System;
using SlamHip;

namespace CoreSLAM
{
    public class ObstacleMap
    {
        private readonly Handle cs;

        /// <summary>Host mirror, indexed [y, x] (ObstacleMap.cs:31); row-major in memory like the device copy.</summary>
        public readonly sbyte[,] Pixels;

        public int Size { get; }

        public float Scale { get; }

        internal ObstacleMap(Handle cs, int sizePixels, float scale)
        {
            this.cs = cs;
            Size = sizePixels;
            Scale = scale;
            Pixels = new sbyte[sizePixels, sizePixels];
        }

        public unsafe void Download()
        {
            fixed (sbyte* p = Pixels)
                Native.Check(Native.slamhip_cs_obstaclemap_download(cs.Ptr, p, (nuint)Pixels.Length));
        }

        public unsafe void Upload()
        {
            fixed (sbyte* p = Pixels)
                Native.Check(Native.slamhip_cs_obstaclemap_upload(cs.Ptr, p, (nuint)Pixels.Length));
        }
    }
}
