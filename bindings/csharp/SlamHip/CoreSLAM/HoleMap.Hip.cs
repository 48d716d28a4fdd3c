// CoreSLAM.HoleMap on the GPU: the public surface of the reference class (CoreSLAM/HoleMap.cs:10-56) with the pixels
// living in device memory.  `Pixels` stays a managed ushort[] because callers read it directly
// (Simulation/MainWindow.xaml.cs:229): it is a MIRROR, refreshed by Download() (everything), Mirror() (blocking: the rectangle
// the scans since the last refresh touched) or MirrorAsync() (the 16-byte units that changed, pushed from a copy stream while
// the next scan runs).  By default (MirrorMode.OnRead) the `Pixels` getter itself asks for what changed since its last read and
// waits for it; with MirrorMode.EveryScan CoreSLAMProcessor.Update issues MirrorAsync() after every scan and the getter only waits
// for the push in flight.  A reader never sees a half-written or outdated map, and a caller that does not look at the map pays
// nothing for it.
using System;
using SlamHip;

namespace CoreSLAM
{
    public class HoleMap
    {
        private readonly Handle cs;                                     // slamhip_cs of the owning processor

        private readonly ushort[] pixels;
        private bool pushInFlight;
        private bool stale = true;                                      // the device map has moved on since the last refresh
        internal MirrorMode Mode = MirrorMode.OnRead;

        internal void MarkStale() { stale = true; }

        /// <summary>Host mirror of the device pixels, row-major [y * Size + x] (HoleMap.cs:27).  Reading it waits for an
        /// asynchronous refresh that is still in flight (slamhip_cs_holemap_mirror_wait).</summary>
        public ushort[] Pixels
        {
            get
            {
                if (stale && Mode == MirrorMode.OnRead && !pushInFlight) MirrorAsync();
                WaitMirror();
                return pixels;
            }
        }

        /// <summary>Side length in pixels (HoleMap.cs:32).</summary>
        public int Size { get; }

        /// <summary>Pixels per metre (HoleMap.cs:37): sizePixels / sizeMeters.</summary>
        public float Scale { get; }

        internal HoleMap(Handle cs, int sizePixels, float scale)
        {
            this.cs = cs;
            Size = sizePixels;
            Scale = scale;
            // (on the pinned object heap: the address never changes while a refresh is in flight.  A managed array does not own its
            // pages, so the library serves it through its pinned staging buffer and copies the changed row ranges in
            // slamhip_cs_holemap_mirror_wait; a host that wants the device to write the mirror directly hands the library
            // page-aligned native memory -- NativeMemory.AlignedAlloc(bytes, 4096) -- and wraps it in a Span)
            pixels = GC.AllocateArray<ushort>(sizePixels * sizePixels, pinned: true);
        }

        /// <summary>Refresh Pixels from the device (2 bytes per pixel over PCIe: 8 MiB at 2048 x 2048).</summary>
        public unsafe void Download()
        {
            WaitMirror();
            fixed (ushort* p = pixels)
                Native.Check(Native.slamhip_cs_holemap_download(cs.Ptr, p, (nuint)pixels.Length));
            stale = false;
        }

        /// <summary>Bring Pixels up to date by copying only what the updates since the last Mirror()/Download() call can have
        /// changed: the bounding rectangle of the scans drawn (the update kernel keeps it on the device).  The whole map on the
        /// first call and after Reset/Upload.</summary>
        public unsafe void Mirror()
        {
            int* rect = stackalloc int[4];
            WaitMirror();
            fixed (ushort* p = pixels)
                Native.Check(Native.slamhip_cs_holemap_mirror(cs.Ptr, p, (nuint)pixels.Length, rect));
            stale = false;
        }

        /// <summary>Asynchronous refresh: enqueues the snapshot of what changed since the last refresh behind the map updates in
        /// flight and returns at once; the data arrive from a copy stream while the next scan runs.  (The pinned array is
        /// page-locked and mapped into the device's address space by the library on the first call.)</summary>
        public unsafe void MirrorAsync()
        {
            fixed (ushort* p = pixels)
                Native.Check(Native.slamhip_cs_holemap_mirror_async(cs.Ptr, p, (nuint)pixels.Length));
            pushInFlight = true;
            stale = false;                                              // (everything up to this request is on its way)
        }

        /// <summary>Waits for the refresh in flight, if any (the `Pixels` getter calls it).</summary>
        public unsafe void WaitMirror()
        {
            if (!pushInFlight) return;
            long px;
            Native.Check(Native.slamhip_cs_holemap_mirror_wait(cs.Ptr, null, &px));
            pushInFlight = false;
        }

        /// <summary>Replace the device pixels with Pixels (restoring a saved map).</summary>
        public unsafe void Upload()
        {
            WaitMirror();
            fixed (ushort* p = pixels)
                Native.Check(Native.slamhip_cs_holemap_upload(cs.Ptr, p, (nuint)pixels.Length));
        }

        /// <summary>Two pixels per byte, the 4 most significant bits of each (HoleMap.cs:44-55) -- packed on the device, so only
        /// a quarter of the map crosses PCIe.</summary>
        public unsafe byte[] GetPackedPixels()
        {
            byte[] packed = new byte[pixels.Length / 2];
            fixed (byte* p = packed)
                Native.Check(Native.slamhip_cs_holemap_download_packed(cs.Ptr, p, (nuint)packed.Length));
            return packed;
        }
    }
}
