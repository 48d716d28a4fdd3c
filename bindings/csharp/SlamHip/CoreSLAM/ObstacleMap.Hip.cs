// CoreSLAM.ObstacleMap on the GPU (reference: CoreSLAM/ObstacleMap.cs:10-42): negative = unmapped, 0 = clear,
// positive = hits.  `Pixels` is the managed sbyte[y, x] mirror of the device map.
using System;
using SlamHip;

namespace CoreSLAM
{
    public class ObstacleMap
    {
        private readonly Handle cs;

        private readonly sbyte[,] pixels;
        private bool stale = true;                                      // the device map has moved on since the last download
        internal MirrorMode Mode = MirrorMode.OnRead;

        internal void MarkStale() { stale = true; }

        /// <summary>Host mirror, indexed [y, x] (ObstacleMap.cs:31: a readonly field there, a property here -- reads compile
        /// unchanged); row-major in memory like the device copy.  Unless the processor's MirrorMode is Manual, a read after the
        /// device map has moved on downloads it first (one byte per pixel: 256 KB at 512 x 512).</summary>
        public sbyte[,] Pixels
        {
            get { if (stale && Mode != MirrorMode.Manual) Download(); return pixels; }
        }

        public int Size { get; }

        public float Scale { get; }

        internal ObstacleMap(Handle cs, int sizePixels, float scale)
        {
            this.cs = cs;
            Size = sizePixels;
            Scale = scale;
            pixels = new sbyte[sizePixels, sizePixels];
        }

        public unsafe void Download()
        {
            fixed (sbyte* p = pixels)
                Native.Check(Native.slamhip_cs_obstaclemap_download(cs.Ptr, p, (nuint)pixels.Length));
            stale = false;
        }

        public unsafe void Upload()
        {
            fixed (sbyte* p = pixels)
                Native.Check(Native.slamhip_cs_obstaclemap_upload(cs.Ptr, p, (nuint)pixels.Length));
        }
    }
}
